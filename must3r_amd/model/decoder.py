"""``MUSt3R`` -- drop-in for must3r/model/decoder.py:14 backed by libmust3r_hip.

Same constructor arguments, attributes (``pointmaps_activation``, ``memory_mode``,
``change_memory_mode``, ``depth``, ``embed_dim``), state-dict keys and
``forward(x, pos, true_shape, current_mem=None, render=False) -> (mem_tuple, pointmaps)`` contract with
tensor or list inputs (decoder.py:158-350).  One forward = ONE native call (``must3r_hip_decode``).

Memory tuple (decoder.py:337): ``(list[depth] of [1,Nm,mem_D], labels int64[1,Nm], n_imgs,
n_protected_imgs, n_protected_tokens)``.  The value tensors are plain torch tensors the caller may index,
overwrite or rebuild (engine/inference.py:205-228).  MI355X-first difference: they are prefix views of
over-allocated per-layer buffers, so an update appends K|V rows in place instead of re-concatenating the
whole memory every call (decoder.py:239/330 is O(Nm) HBM traffic per layer per call).  Appending is only
done when the caller hands back exactly the newest view; anything else (boolean-indexed copies, an older
tuple, another dtype) is copied into a fresh buffer first, so aliasing is never observable.

Precision: operands follow the active ``torch.autocast`` dtype like the reference's Linear layers
(demo/inference.py:198); without autocast ``self.precision`` is used.  Residual stream, LayerNorm,
softmax, accumulators and the prediction head (decoder.py:152-153 forces fp32) are fp32 / fp32-equivalent.
"""
import ctypes as C
from functools import partial

import torch
import torch.nn as nn

from .. import _lib
from ..config import ModelConfig
from .blocks import (ActivationType, DecBlockParams, LinearHeadParams, MlpParams, MEMORY_MODES, init_weights,
                     parse_pos_embed)
from ._hip_module import HipModule, operand_dtype, autocast_dtype, _DT, _TORCH_DT

_MEM_MODE = {"kv": _lib.MEM_KV, "norm_y": _lib.MEM_NORM_Y, "raw": _lib.MEM_RAW}


class _MemBuffers:
    """Per-layer over-allocated K|V buffers ``[B, cap, mem_D]`` (scene b's rows at ``b * cap``); ``valid`` = rows handed out
    by the newest view."""

    def __init__(self, depth, cap, mem_D, dtype, device, batch=1):
        self.bufs = [torch.empty((batch, cap, mem_D), dtype=dtype, device=device) for _ in range(depth)]
        self.cap = cap
        self.batch = batch
        self.valid = 0

    def views(self, n):
        out = []
        for b in self.bufs:
            v = b[:, :n]
            v._m3r_owner = self
            out.append(v)
        return out


class MUSt3R(HipModule):
    _part = _lib.PART_DECODER
    _prefix = "decoder."

    def __init__(self, img_size=(224, 224), enc_embed_dim=1024, patch_size=16, embed_dim=768, output_dim=1792,
                 depth=12, num_heads=12, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU,
                 pos_embed="RoPE100", landscape_only=True, head="Linear", feedback_type=None, memory_mode="norm_y",
                 pointmaps_activation=ActivationType.NORM_EXP, block_type=None, precision="fp16wa", **kv):
        super().__init__()
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        assert memory_mode in MEMORY_MODES
        if head != "Linear":
            raise ValueError(f"invalid head {head}")  # decoder.py:80
        if feedback_type not in ("single_mlp", "single_linear", None, "", False):
            raise ValueError(f"Unknown {feedback_type=}")   # feedback_mechanism.py:18-19 asserts on anything else
        assert output_dim == patch_size * patch_size * 7
        self.pointmaps_activation = pointmaps_activation
        self.depth = depth
        self.embed_dim = embed_dim
        self.memory_mode = memory_mode
        self.attn_num_heads = num_heads
        self.feedback_type = feedback_type
        self.reserve_memory_tokens = 0   # capacity hint for freshly allocated memory buffers (see _writable_memory)
        self.landscape_only = landscape_only
        self.max_seq_len = max(img_size) // patch_size
        self.grid_size = (img_size[0] // patch_size, img_size[1] // patch_size)
        freq, f0 = parse_pos_embed(pos_embed)
        # parameters, reference attribute names (decoder.py:49-83)
        self.feat_embed_enc_to_dec = nn.Linear(enc_embed_dim, embed_dim, bias=True)
        self.image2_embed = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.blocks_dec = nn.ModuleList([DecBlockParams(embed_dim, mlp_ratio, memory_mode) for _ in range(depth)])
        # feedback_mechanism.py:11-22 create_feedback_layers
        if feedback_type == "single_mlp":
            self.feedback_layer = MlpParams(embed_dim, 4 * embed_dim, embed_dim)
            self.feedback_norm = nn.LayerNorm(embed_dim)  # default eps 1e-5 (feedback_mechanism.py:14)
        elif feedback_type == "single_linear":
            self.feedback_layer = nn.Linear(embed_dim, embed_dim)
            self.feedback_norm = nn.LayerNorm(embed_dim)
        else:
            self.feedback_layer = None
            self.feedback_norm = None
        self.norm_dec = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head_dec = LinearHeadParams(embed_dim, output_dim, patch_size)
        init_weights(self)
        torch.nn.init.normal_(self.image2_embed, std=0.02)
        if feedback_type == "single_mlp":                      # feedback_mechanism.py:26-36: inactive at the start
            nn.init.constant_(self.feedback_layer.fc2.bias, 0)
            nn.init.constant_(self.feedback_layer.fc2.weight, 0)
        elif feedback_type == "single_linear":
            nn.init.constant_(self.feedback_layer.bias, 0)
            nn.init.constant_(self.feedback_layer.weight, 0)
        self._hip_init(ModelConfig(img_size=max(img_size), patch_size=patch_size, enc_dim=enc_embed_dim,
                                   enc_heads=enc_embed_dim // 64, dec_dim=embed_dim, dec_depth=depth, dec_heads=num_heads,
                                   mlp_ratio=mlp_ratio, rope_freq=freq, rope_f0=f0), precision)

    # -- reference API -------------------------------------------------------------------------
    def change_memory_mode(self, memory_mode="norm_y"):  # decoder.py:113-117
        assert memory_mode in MEMORY_MODES
        for blk in self.blocks_dec:
            blk.memory_mode = memory_mode
        self.memory_mode = memory_mode

    def from_dust3r(self, state_dict, verbose=True, load_head=False):  # decoder.py:85-94
        state_dict = {k.replace("dec_blocks.", "blocks_dec.").replace("decoder_embed.", "feat_embed_enc_to_dec.").replace(
            "dec_norm.", "norm_dec."): v for k, v in state_dict.items()}
        if load_head:
            state_dict = {k.replace("downstream_head.proj.", "head_dec.proj."): v for k, v in state_dict.items()}
        inc = self.load_state_dict(state_dict, strict=False)
        if verbose:
            print(inc)
        return inc

    def from_croco(self, state_dict, verbose=True):
        return self.from_dust3r(state_dict, verbose=verbose)

    def _operand(self):
        ac = autocast_dtype()
        if ac == torch.bfloat16:
            return _lib.BF16
        if ac == torch.float16 and self.precision == "bf16":
            return _lib.F16
        return operand_dtype(self.precision)  # no autocast, or fp16 autocast with an fp16-family precision

    # -- memory management ---------------------------------------------------------------------
    def _writable_memory(self, mem_vals, Nm, R, tdt, device, B=1, mem_D=None):
        """Return the buffers whose rows [0,Nm) (of every scene) hold ``mem_vals`` and that can take R more rows per scene."""
        if mem_D is None:
            mem_D = 2 * self.embed_dim if self.memory_mode == "kv" else self.embed_dim
        owner = None
        if mem_vals is not None and len(mem_vals) == self.depth:
            owner = getattr(mem_vals[0], "_m3r_owner", None)
            ok = owner is not None and owner.valid == Nm and owner.cap >= Nm + R and owner.batch == B
            if ok:
                for v, b in zip(mem_vals, owner.bufs):
                    if getattr(v, "_m3r_owner", None) is not owner or v.data_ptr() != b.data_ptr() or v.dtype != tdt \
                            or v.shape[0] != B or v.shape[1] != Nm or v.shape[2] != mem_D or v.device != b.device:
                        ok = False
                        break
            if not ok:
                owner = None
        if owner is None:
            # growth doubles the capacity (amortised O(1) appends); a caller that knows how many tokens the memory will
            # reach (engine.run_scene: keyframes x tokens) can say so through ``reserve_memory_tokens`` and skip the
            # 12 x log2 re-allocation copies of a scene (measured: 49 copies, 0.5 ms per 20-view scene)
            # The hint is ONE-SHOT: it sizes the first buffer of a scene (Nm == 0) and is consumed there, so later fresh
            # allocations (copy-out after a discarded / older tuple, another scene) are not sized for it.
            hint = int(getattr(self, "reserve_memory_tokens", 0) or 0) if Nm == 0 else 0
            if Nm == 0:
                self.reserve_memory_tokens = 0
            cap = max(Nm + R, 2 * Nm, 1024, hint)
            owner = _MemBuffers(self.depth, cap, mem_D, tdt, device, B)
            if Nm > 0:
                for v, b in zip(mem_vals, owner.bufs):
                    b[:, :Nm].copy_(v.reshape(B, Nm, mem_D))
        return owner

    def _check_memory_rows(self, mem_vals, mem_D, fp8_rows):
        """The row format of a memory handed back by the caller must be the one this module's mode reads: fp8 'kv' rows are opaque
        bytes [K e4m3: D | V 16-bit: 2 D] (uint8, 3 D per row), every other mode holds 16-bit (or wider floating) elements, mem_D per row.  A
        numeric cast between the two would hand the kernels garbage and, in one direction, let them read past the buffer -- refuse instead."""
        for v in mem_vals:
            if fp8_rows:
                if v.dtype != torch.uint8 or int(v.shape[2]) != mem_D:
                    raise ValueError(f"attention_fp8 with memory_mode 'kv': the memory must be the uint8 [B, Nm, {mem_D}] rows an fp8-mode "
                                     f"update returned, got {v.dtype} [.., {int(v.shape[2])}] (a memory written with attention_fp8 off cannot "
                                     "be read with it on; re-run the update in this mode)")
            elif not v.dtype.is_floating_point or int(v.shape[2]) != mem_D:
                raise ValueError(f"memory_mode {self.memory_mode!r} without attention_fp8: the memory must be floating-point [B, Nm, {mem_D}] rows, "
                                 f"got {v.dtype} [.., {int(v.shape[2])}] (a memory written with attention_fp8 on holds opaque e4m3 | 16-bit bytes "
                                 "and cannot be read with it off; re-run the update in this mode)")

    # -- forward -------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, pos, true_shape, current_mem=None, render=False, return_feats=False, pointmaps_out=None, cp=None):
        """``pointmaps_out`` (extension, r05): where to write the raw pointmaps -- a fp32 cuda tensor [B, n, H, W, 7] (a list of them for list inputs) whose last four
        dimensions are contiguous; its batch stride is free, so a caller walking a scene's views over several calls can pass the slice ``buf[:, i:i+n]`` of one
        [B, V, H, W, 7] buffer (engine.run_scenes) instead of concatenating the calls' outputs.  Returned in place of a fresh tensor.

        ``cp`` (extension, r06): a ``must3r_amd.parallel.ContextParallel`` -- ``current_mem`` is THIS rank's shard of a memory spread over the ranks of ``cp.group``
        (``cp.n_mem_total`` rows in all); the cross attention of the call attends the local rows, exchanges one fp32 partial per layer with the other ranks
        (all-gather) and merges them (include/must3r_hip.h ``must3r_hip_cp``).  One-view memory updates of one scene in memory_mode 'kv' only."""
        is_list = isinstance(x, (list, tuple))
        if is_list:
            return_feats = False   # the reference's list dispatch drops the flag (decoder.py:270); forward_list honours it
        return self._forward(x, pos, true_shape, current_mem, render, return_feats, pointmaps_out, cp)

    def _forward(self, x, pos, true_shape, current_mem, render, return_feats, pointmaps_out=None, cp=None):
        is_list = isinstance(x, (list, tuple))
        xs = list(x) if is_list else [x]
        poss = list(pos) if is_list else [pos]
        shapes = list(true_shape) if is_list else [true_shape]
        B = int(xs[0].shape[0])
        assert all(int(t.shape[0]) == B for t in xs + poss), "all groups must share the batch size"
        # B > 1 (the reference's batch dimension, decoder.py:170-186): the scenes of a batch never interact -- memory, labels and
        # attention are all per batch element -- and are decoded by ONE native call (must3r_hip_decode_args.n_scenes): every GEMM
        # sees M = B x rows, the attention tables B x views, each scene its own rows of the [B, capacity, mem_D] memory buffers.
        pouts = None if pointmaps_out is None else (list(pointmaps_out) if isinstance(pointmaps_out, (list, tuple)) else [pointmaps_out])
        out, outs, feats = self._forward_scene(xs, poss, shapes, current_mem, render, return_feats, pouts, cp)
        if return_feats:
            return out, (outs if is_list else outs[0]), (feats if is_list else feats[0])
        return out, (outs if is_list else outs[0])

    def _forward_scene(self, xs, poss, shapes, current_mem, render, return_feats, pouts=None, cp=None):
        """B scenes of identical shapes: ONE native decode call.  Returns (memory, [pointmaps per group], [feats per group] | None)."""
        # (The per-call view tables travel through a 64 KiB staging slot, 1365 views; the reference renders every view of an aspect
        # ratio in ONE call when the caller sets no max_bs (engine/inference.py:489-522), so the LIBRARY cuts larger render calls
        # into ranges of views / scenes: rendered views are independent.  Only return_feats needs the cut up here, for its buffer.)
        MAXV = 1024
        B = int(xs[0].shape[0])
        if render and return_feats and B == 1 and sum(int(x.shape[1]) for x in xs) > MAXV:
            outs, feats = [], []
            for xi, pi, ti in zip(xs, poss, shapes):
                pms, fts = [], []
                tsi = ti.reshape(1, -1, 2)
                for a in range(0, int(xi.shape[1]), MAXV):
                    _, o, f = self._forward_scene([xi[:, a:a + MAXV]], [pi[:, a:a + MAXV]], [tsi[:, a:a + MAXV]], current_mem, True,
                                                  return_feats)
                    pms.append(o[0])
                    fts.append(f[0] if f is not None else None)
                outs.append(torch.cat(pms, dim=1))
                feats.append([torch.cat([f[l] for f in fts], dim=1) for l in range(len(fts[0]))] if return_feats else None)
            return current_mem, outs, (feats if return_feats else None)
        ctx = self._context()
        dev = self._ctx_dev
        device = torch.device("cuda", dev)
        odt = self._operand()
        tdt = _TORCH_DT[odt]
        fp8_rows = self.attention_fp8 and self.memory_mode == "kv"
        if fp8_rows:
            tdt = torch.uint8   # opaque rows [K e4m3: D bytes | V 16-bit: 2 D bytes] = 3/4 of the 16-bit footprint (include/must3r_hip.h)
        D = self.embed_dim
        assert not render or current_mem is not None  # decoder.py:278

        groups = (_lib.Group * len(xs))()
        keep = []
        outs = []
        R = 0
        nimgs, Ns = [], []
        for i, (xi, pi, ti) in enumerate(zip(xs, poss, shapes)):
            _, n, N, Cenc = xi.shape
            xi = self._check_input(xi, "x", torch.float32)
            pi = self._check_input(pi, "pos", torch.int64)
            ts = ti.reshape(-1, 2)
            if ts.is_cuda:  # device->host sync, like the reference's head wrapper (head.py:33); pass it on the host to avoid
                ts = ts.cpu()
            assert bool((ts[0:1] == ts).all()), "true_shape must be all identical"  # head.py:31
            H, W = (int(v) for v in ts[0].tolist())
            assert (H // 16) * (W // 16) == N, (H, W, N)
            sstride = 0
            if pouts is not None:
                pm = pouts[i]
                inner = n * H * W * 7
                if (not pm.is_cuda or pm.dtype != torch.float32 or tuple(pm.shape) != (B, n, H, W, 7) or pm.device != device
                        or tuple(pm.stride()[1:]) != (H * W * 7, W * 7, 7, 1) or (B > 1 and (pm.stride(0) < inner or pm.stride(0) % 4))
                        or pm.data_ptr() % 16):   # (the head epilogue stores 16-byte vectors: scene stride % 4 floats, 16-byte-aligned base)
                    raise ValueError(f"pointmaps_out[{i}]: need a fp32 cuda tensor [{B}, {n}, {H}, {W}, 7] with contiguous views, got {tuple(pm.shape)} "
                                     f"{pm.dtype} strides {tuple(pm.stride())}")
                sstride = int(pm.stride(0)) if B > 1 and int(pm.stride(0)) != inner else 0
            else:
                pm = torch.empty((B, n, H, W, 7), dtype=torch.float32, device=device)
            keep += [xi, pi]
            outs.append(pm)
            groups[i] = _lib.Group(xi.data_ptr(), pi.data_ptr(), n, N, H, W, pm.data_ptr(), sstride)
            R += n * N
            nimgs.append(n)
            Ns.append(N)

        if current_mem is None:
            mem_vals, mem_labels, mem_nimgs, mem_prot_imgs, mem_prot_tok = None, torch.zeros((B, 0), dtype=torch.int64,
                                                                                            device=device), 0, 0, 0
            Nm = 0
        else:
            mem_vals, mem_labels, mem_nimgs, mem_prot_imgs, mem_prot_tok = current_mem
            Nm = int(mem_vals[0].shape[1])
            assert all(int(v.shape[0]) == B for v in mem_vals), "memory and inputs must share the batch size"

        mem_D = (3 * D if fp8_rows else 2 * D) if self.memory_mode == "kv" else D
        if current_mem is not None:
            self._check_memory_rows(mem_vals, mem_D, fp8_rows)
        if render:
            # read-only: any [B, Nm, mem_D] tensors whose rows are contiguous and whose scene stride is the same in every layer
            # (the prefix views of this module's own buffers are: stride cap x mem_D) are read in place
            vals, stride = [], None
            for v in mem_vals:
                if not v.is_cuda or v.dtype != tdt or v.stride(2) != 1 or v.stride(1) != mem_D or (B > 1 and v.stride(0) % mem_D):
                    v = v.to(device=device, dtype=tdt).contiguous()
                vals.append(v)
            if B > 1:
                strides = {int(v.stride(0)) // mem_D for v in vals}
                if len(strides) != 1 or min(strides) < Nm:
                    vals = [v.contiguous() for v in vals]
                stride = int(vals[0].stride(0)) // mem_D
            ptrs = (C.c_void_p * self.depth)(*[v.data_ptr() for v in vals])
            keep += vals
            cap, stride = Nm, (0 if B == 1 else stride)   # read-only: the rows that exist are the capacity
        else:
            owner = self._writable_memory(mem_vals, Nm, R, tdt, device, B, mem_D)
            ptrs = (C.c_void_p * self.depth)(*[b.data_ptr() for b in owner.bufs])
            cap, stride = owner.cap, owner.cap

        # return_feats (decoder.py:344-347): [encoder tokens, residual stream after blocks 0..depth-2, norm_dec(last)] -- fp32
        # here (the residual stream is fp32 on this path, bf16 in the reference under autocast)
        feats_buf = torch.empty((self.depth, B, R, D), dtype=torch.float32, device=device) if return_feats else None
        cp_struct = None
        if cp is not None:
            if render or current_mem is None or B != 1 or len(xs) != 1 or nimgs[0] != 1 or self.memory_mode != "kv" or self.attention_fp8:
                raise ValueError("context-parallel cross attention is for one-view memory updates of one scene against an existing memory, memory_mode 'kv', "
                                 "16-bit attention operands")
            cp_struct = cp.native_args(ctx, R, device)   # allocates / reuses the slot buffer; keeps the callback alive
        args = _lib.DecodeArgs(odt | (_lib.ATTN_FP8 if self.attention_fp8 else 0), _MEM_MODE[self.memory_mode],
                               1 if render else 0, 1 if current_mem is None else 0,
                               len(xs), groups, Nm, ptrs, feats_buf.data_ptr() if return_feats else None,
                               cap, B, stride, C.pointer(cp_struct) if cp_struct is not None else None, 1 if self._causal else 0)
        rc = ctx.lib.must3r_hip_decode(ctx.handle, C.byref(args), self._stream(dev))
        if cp is not None and rc != 0:
            cp.reraise()          # an exception raised inside the exchange callback (ctypes cannot propagate it) comes out here
        _lib.check(rc)

        if render:
            out = current_mem  # decoder.py:252 / :339: the memory comes back untouched
        else:
            owner.valid = Nm + R
            new_vals = owner.views(Nm + R)
            labels = []
            k = 0
            # host mirror of the label layout (engine._label_runs): lets the L3 memory surgery build its indices without
            # reading the labels back from the device
            from ..engine import _label_runs, attach_label_runs
            runs = [] if Nm == 0 else _label_runs(mem_labels)    # None: no mirror, or the labels were edited in place since it was attached
            runs = list(runs) if runs is not None and sum(c for _, c in runs) == Nm else None
            for n, N in zip(nimgs, Ns):  # decoder.py:241-249 / :332-334
                labels.append((torch.arange(n, dtype=torch.int64, device=device) + (mem_nimgs + k)).repeat_interleave(N).view(1, -1)
                              .expand(B, -1))
                if runs is not None:
                    runs += [(mem_nimgs + k + j, N) for j in range(n)]
                k += n
            mem_labels = torch.cat([mem_labels.to(device)] + labels, dim=1)
            if runs is not None and B == 1:
                attach_label_runs(mem_labels, runs)
            tot = mem_nimgs + sum(nimgs)
            out = self._memory_tail(new_vals, mem_labels, tot, mem_prot_imgs, mem_prot_tok, sum(nimgs), Ns[0])
        feats = None
        if return_feats:
            feats, r0 = [], 0
            for xi, n, N in zip(xs, nimgs, Ns):
                feats.append([xi] + [feats_buf[l, :, r0:r0 + n * N].reshape(B, n, N, D) for l in range(self.depth)])
                r0 += n * N
        return out, outs, feats

    def forward_list(self, x, pos, true_shape, current_mem=None, render=False, return_feats=False):  # decoder.py:158
        return self._forward(list(x), list(pos), list(true_shape), current_mem, render, return_feats)

    # MUSt3R: every image is "protected" (decoder.py:336: (mem, labels, n, n, Nm')); CausalMUSt3R overrides both
    _causal = False

    def _memory_tail(self, vals, labels, n_imgs, prot_imgs, prot_tok, n_new, N):
        return (vals, labels, n_imgs, n_imgs, labels.shape[1])


class CausalMUSt3R(MUSt3R):
    """The class the reference's checkpoints are TRAINED as (decoder.py:352-553): ONE forward over a sequence of views in which view i cross-attends the memory of the
    views before it.  Built in r06 for SURVEY.md section 8(f) "later" as a FORWARD (what a function of its inputs can be checked against): the memory dropout of the
    training recipe (``mem_dropout > 0``: random token selection, decoder.py:470-485) is refused.  ``use_mem_mask`` / ``use_xformers_mask`` choose between two
    equivalent ways of masking in the reference (physically removed rows / an additive mask) and change nothing here: the native cross attention reads a key PREFIX per
    view (``must3r_hip_decode_args.causal``).  Tensor inputs only, like the reference's class (its forward has no list dispatch).  Render calls and one-view updates are
    MUSt3R's; the tuple's tail follows decoder.py:461-464 (``protected_imgs``)."""

    _causal = True

    def __init__(self, protected_imgs=1, mem_dropout=0.0, dropout_mode="temporary", use_xformers_mask=False, use_mem_mask=False, **kv):
        if dropout_mode not in ("temporary", "permanent"):
            raise ValueError(f"Invalid dropout mode = {dropout_mode}")            # decoder.py:376
        if float(mem_dropout) > 0.0:
            raise NotImplementedError("CausalMUSt3R(mem_dropout > 0) is the training-time random token dropout (decoder.py:470-485): not part of the forward path")
        super().__init__(**kv)
        self.protected_imgs = int(protected_imgs)
        self.dropout_mode = dropout_mode
        self.use_xformers_mask, self.use_mem_mask = bool(use_xformers_mask), bool(use_mem_mask)

    @torch.no_grad()
    def forward(self, x, pos, true_shape, current_mem=None, render=False, return_feats=False, pointmaps_out=None):
        if isinstance(x, (list, tuple)):
            raise TypeError("CausalMUSt3R.forward takes tensors [B, nimgs, N, C] (the reference's class has no list dispatch, decoder.py:435-439)")
        return self._forward(x, pos, true_shape, current_mem, render, return_feats, pointmaps_out)

    def forward_list(self, *a, **k):
        raise TypeError("CausalMUSt3R has no forward_list (decoder.py:352-553)")

    def _memory_tail(self, vals, labels, n_imgs, prot_imgs, prot_tok, n_new, N):
        prot = min(self.protected_imgs, int(prot_imgs) + int(n_new))             # decoder.py:461-464
        return (vals, labels, n_imgs, prot, int(prot_tok) + (prot - int(prot_imgs)) * int(N))
