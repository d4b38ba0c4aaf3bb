"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the MUSt3R multi-view forward path.

A functional fp32 restatement (torch CPU tensors, state-dict in / tensors out) of the reference
algorithm on the hot path named by BASELINE.json.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file; the product package ``must3r_amd`` never
does (its HIP path fails loudly when the extension is missing instead of falling back here).

Every function cites the reference lines it follows (paths relative to /root/reference).

Pinning status: the reference has no tests or golden vectors.  This
restatement is pinned instead against the reference's *own code* run in the build container
(``oracle/ref_shims.py`` + ``oracle/make_golden.py`` -> ``tests/golden/*.npz``; the cross-check
``tests/test_oracle_vs_reference.py`` re-runs the real reference whenever /root/reference exists).
The un-vendored leaves (croco ``Mlp``/``RoPE2D``/``PositionGetter``, dust3r ``PatchEmbedDust3R``,
softmax attention) are restated from SURVEY.md Appendix A: for those leaves parity is UNPINNED
(no upstream source or vector is available offline).

``opq`` (operand quantiser) is an optional hook applied to every matrix-multiply operand; passing
``lambda t: t.bfloat16().float()`` models the HIP path's precision policy (bf16 MFMA operands, fp32
accumulation / residual stream / LayerNorm / softmax) so tests can derive kernel-level tolerances.
"""
import math

import torch
import torch.nn.functional as F

_ID = (lambda t: t)


# ------------------------------------------------------------------------------------------------
# leaves
# ------------------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps):
    """nn.LayerNorm over the last dim (eps 1e-6: encoder.py:21, decoder.py:28; 1e-5 for
    feedback_norm: feedback_mechanism.py:14)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * w + b


def linear(x, w, b, opq=_ID):
    """nn.Linear: x @ w.T + b."""
    y = opq(x) @ opq(w).t()
    return y + b if b is not None else y


def gelu(x):
    """nn.GELU() default = exact erf form (croco Mlp)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def mlp(sd, p, x, opq=_ID):
    """croco.models.blocks.Mlp: fc2(GELU(fc1(x)))  (used at layers.py:48,78; feedback_mechanism.py:13)."""
    h = gelu(linear(x, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"], opq))
    return linear(h, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"], opq)


def rope_tables(npos, freq=100.0, f0=1.0, half=32):
    """cos/sin[npos, half/2] of angle p * f0 * freq^(-i/(half/2)), i in [0, half/2)."""
    i = torch.arange(0, half, 2, dtype=torch.float32) / half
    inv = f0 / (freq ** i)
    ang = torch.outer(torch.arange(npos, dtype=torch.float32), inv)
    return ang.cos(), ang.sin()


def rope2d(t, pos, freq=100.0, f0=1.0):
    """croco RoPE2D on t[B,H,N,64], pos[B,N,2] (applied at attention.py:42-44).

    dims [0,32) rotate with y = pos[...,0], dims [32,64) with x = pos[...,1]; inside a half the
    rotate-half pairs are (i, i+16), 16 frequencies 100^(-i/16)."""
    B, H, N, D = t.shape
    half = D // 2
    q = half // 2
    cos, sin = rope_tables(int(pos.max()) + 1, freq, f0, half)
    out = torch.empty_like(t)
    for h0, axis in ((0, 0), (half, 1)):
        c = cos[pos[:, :, axis]][:, None]  # [B,1,N,q]
        s = sin[pos[:, :, axis]][:, None]
        a = t[..., h0:h0 + q]
        b = t[..., h0 + q:h0 + half]
        out[..., h0:h0 + q] = a * c - b * s
        out[..., h0 + q:h0 + half] = b * c + a * s
    return out


def softmax_attention(q, k, v, opq=_ID, sdpa=False):
    """softmax(q k^T / sqrt(64)) v on [B,H,N,64]  (attention.py:75-78; ==SDPA :70 == xformers :63)."""
    if sdpa and opq is _ID:
        return F.scaled_dot_product_attention(q, k, v)
    s = (opq(q) @ opq(k).transpose(-2, -1)) * (q.shape[-1] ** -0.5)
    p = torch.softmax(s, dim=-1)
    return opq(p) @ opq(v)


def split_heads(x, H):
    B, N, C = x.shape
    return x.view(B, N, H, C // H).permute(0, 2, 1, 3)


def merge_heads(x):
    B, H, N, D = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, N, H * D)


def self_attention(sd, p, x, pos, H, cfg, opq=_ID, sdpa=False):
    """Attention.forward attention.py:92-99 (+ RoPE on q,k :42-44)."""
    B, N, C = x.shape
    qkv = linear(x, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"], opq)
    q, k, v = (split_heads(t, H) for t in qkv.view(B, N, 3, C).unbind(2))
    q = rope2d(q, pos, cfg.rope_freq, cfg.rope_f0)
    k = rope2d(k, pos, cfg.rope_freq, cfg.rope_f0)
    o = merge_heads(softmax_attention(q, k, v, opq, sdpa))
    return linear(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"], opq)


def patch_embed(sd, img, patch):
    """dust3r PatchEmbedDust3R: Conv2d(3,C,k=s=16) -> flatten(2).transpose(1,2); positions = row-major
    (y,x) grid (call sites encoder.py:43,48)."""
    B, _, Hh, Ww = img.shape
    assert Hh % patch == 0 and Ww % patch == 0
    gh, gw = Hh // patch, Ww // patch
    x = F.conv2d(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    x = x.flatten(2).transpose(1, 2).contiguous()
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    pos = torch.stack((ys.reshape(-1), xs.reshape(-1)), dim=-1).view(1, gh * gw, 2).expand(B, -1, -1).contiguous()
    return x, pos


# ------------------------------------------------------------------------------------------------
# encoder  (Dust3rEncoder.forward encoder.py:46-52; Block.forward layers.py:51-54)
# ------------------------------------------------------------------------------------------------
def encoder_forward(sd, cfg, img, true_shape=None, opq=_ID, sdpa=False):
    x, pos = patch_embed(sd, img.float(), cfg.patch_size)
    if opq is not _ID:  # the HIP path runs the patch GEMM on bf16 operands as well
        B, _, Hh, Ww = img.shape
        p = cfg.patch_size
        cols = F.unfold(img.float(), kernel_size=p, stride=p).transpose(1, 2)  # [B,N,3*p*p]
        x = linear(cols, sd["patch_embed.proj.weight"].reshape(cfg.enc_dim, -1), sd["patch_embed.proj.bias"], opq)
    for i in range(cfg.enc_depth):
        b = f"blocks_enc.{i}"
        h = layer_norm(x, sd[b + ".norm1.weight"], sd[b + ".norm1.bias"], 1e-6)
        x = x + self_attention(sd, b + ".attn", h, pos, cfg.enc_heads, cfg, opq, sdpa)
        h = layer_norm(x, sd[b + ".norm2.weight"], sd[b + ".norm2.bias"], 1e-6)
        x = x + mlp(sd, b + ".mlp", h, opq)
    x = layer_norm(x, sd["norm_enc.weight"], sd["norm_enc.bias"], 1e-6)
    return x, pos


# ------------------------------------------------------------------------------------------------
# decoder
# ------------------------------------------------------------------------------------------------
def prepare_y(sd, b, y, memory_mode, opq=_ID):
    """CachedDecoderBlock.prepare_y layers.py:81-88."""
    if memory_mode == "raw":
        return y
    yn = layer_norm(y, sd[b + ".norm_y.weight"], sd[b + ".norm_y.bias"], 1e-6)
    if memory_mode == "norm_y":
        return yn
    k = linear(yn, sd[b + ".cross_attn.projk.weight"], sd[b + ".cross_attn.projk.bias"], opq)
    v = linear(yn, sd[b + ".cross_attn.projv.weight"], sd[b + ".cross_attn.projv.bias"], opq)
    return torch.cat((k, v), dim=-1)


def decoder_block(sd, b, x, y, pos, cfg, memory_mode, opq=_ID, sdpa=False):
    """CachedDecoderBlock.forward layers.py:90-99 (+ CachedCrossAttention.forward attention.py:139-149;
    cross-attention has pos_embed=None, layers.py:72 -> no RoPE, no mask)."""
    H, D = cfg.dec_heads, cfg.dec_dim
    h = layer_norm(x, sd[b + ".norm1.weight"], sd[b + ".norm1.bias"], 1e-6)
    x = x + self_attention(sd, b + ".attn", h, pos, H, cfg, opq, sdpa)
    if memory_mode == "kv":
        key, value = y[..., :D], y[..., D:]
    else:
        yn = layer_norm(y, sd[b + ".norm_y.weight"], sd[b + ".norm_y.bias"], 1e-6) if memory_mode == "raw" else y
        key = linear(yn, sd[b + ".cross_attn.projk.weight"], sd[b + ".cross_attn.projk.bias"], opq)
        value = linear(yn, sd[b + ".cross_attn.projv.weight"], sd[b + ".cross_attn.projv.bias"], opq)
    h = layer_norm(x, sd[b + ".norm2.weight"], sd[b + ".norm2.bias"], 1e-6)
    q = split_heads(linear(h, sd[b + ".cross_attn.projq.weight"], sd[b + ".cross_attn.projq.bias"], opq), H)
    o = merge_heads(softmax_attention(q, split_heads(key, H), split_heads(value, H), opq, sdpa))
    x = x + linear(o, sd[b + ".cross_attn.proj.weight"], sd[b + ".cross_attn.proj.bias"], opq)
    h = layer_norm(x, sd[b + ".norm3.weight"], sd[b + ".norm3.bias"], 1e-6)
    return x + mlp(sd, b + ".mlp", h, opq)


def unpatchify_head(sd, cfg, tok, Hh, Ww):
    """_compute_prediction_head decoder.py:149-156 -> LinearHead.forward head.py:67-72 ->
    unpatchify tools/image.py:9-14: feature c*256 + i*16 + j of token (gy,gx) is channel c of pixel
    (16 gy + i, 16 gx + j); output [n,H,W,7].  Always fp32 (autocast off, decoder.py:152-153)."""
    p = cfg.patch_size
    t = layer_norm(tok, sd["norm_dec.weight"], sd["norm_dec.bias"], 1e-6)
    f = linear(t.float(), sd["head_dec.proj.weight"], sd["head_dec.proj.bias"])  # [n,N,7*p*p]
    n = f.shape[0]
    gh, gw = Hh // p, Ww // p
    f = f.view(n, gh, gw, 7, p, p).permute(0, 1, 4, 2, 5, 3)  # n, gy, i, gx, j, c
    return f.reshape(n, Hh, Ww, 7)


def empty_memory(cfg, memory_mode, dtype=torch.float32):
    """MUSt3R._get_empty_memory decoder.py:141-147."""
    mem_D = 2 * cfg.dec_dim if memory_mode == "kv" else cfg.dec_dim
    vals = [torch.zeros((1, 0, mem_D), dtype=dtype) for _ in range(cfg.dec_depth)]
    return vals, torch.zeros((1, 0), dtype=torch.int64), 0, 0, 0


def decoder_forward(sd, cfg, x, pos, true_shape, current_mem=None, render=False, memory_mode="kv",
                    opq=_ID, sdpa=False, return_feats=False, causal=False, protected_imgs=1):
    """MUSt3R.forward / forward_list decoder.py:158-350, batch B = 1.

    ``causal=True``: CausalMUSt3R.forward (decoder.py:435-553) with its memory dropout off (p = 0: the only form that is a function of its inputs): in a memory
    update, view i cross-attends the old memory and the NEW tokens of the views before it in the call (make_attn_mask decoder.py:389-433: labels < idx, != idx) instead
    of the new tokens of all other views; at initialisation view 0 attends view 1's tokens (:399-402).  Tensor inputs only (the class has no list dispatch); the tuple's
    tail is (n_imgs, min(protected_imgs, ...), protected tokens) (:461-464).

    ``x``: tensor [1,n,N,Cenc] or a list of such tensors (one per aspect ratio); ``pos`` / ``true_shape``
    alike.  Returns ``(mem_tuple, pointmaps)`` with the same container type as the input
    (pointmaps [1,n,H,W,7] fp32)."""
    is_list = isinstance(x, (list, tuple))
    xs = list(x) if is_list else [x]
    poss = list(pos) if is_list else [pos]
    shapes = list(true_shape) if is_list else [true_shape]
    D = cfg.dec_dim
    first_call = current_mem is None
    cur = []
    nimgs, Ns = [], []
    for g, xg in enumerate(xs):
        B, n, N, Cenc = xg.shape
        assert B == 1
        t = linear(xg.reshape(n, N, Cenc).float(), sd["feat_embed_enc_to_dec.weight"],
                   sd["feat_embed_enc_to_dec.bias"], opq)  # decoder.py:176 / :275
        e2 = sd["image2_embed"].view(1, 1, D)
        if first_call and g == 0:
            t = torch.cat((t[:1], t[1:] + e2), dim=0)  # decoder.py:177-179 / :280-282: skip the reference view
        else:
            t = t + e2                                   # decoder.py:181 / :287
        cur.append(t)
        nimgs.append(n)
        Ns.append(N)
    if first_call:
        assert not render  # decoder.py:278
        mem_vals, mem_labels, mem_nimgs, mem_prot_imgs, mem_prot_tok = empty_memory(cfg, memory_mode)
    else:
        mem_vals, mem_labels, mem_nimgs, mem_prot_imgs, mem_prot_tok = current_mem
        mem_vals = [m.float() for m in mem_vals]
    Nm = mem_vals[0].shape[1]
    use_mask = (not render) and (Nm > 0 or sum(nimgs) > 1)  # decoder.py:199 / :293
    # token offset of each (group, view) inside the concatenated new tokens (decoder.py:120-131)
    offs = []
    o = 0
    for n, N in zip(nimgs, Ns):
        offs.append([o + j * N for j in range(n)])
        o += n * N
    Nt = o
    new_mem = []
    feats = [[xg.reshape(1, nimgs[g], Ns[g], -1)] for g, xg in enumerate(xs)]  # decoder.py:173 / :274: the encoder tokens
    for l in range(cfg.dec_depth):
        b = f"blocks_dec.{l}"
        if not render:
            x_cat = torch.cat([c.reshape(1, -1, D) for c in cur], dim=1)  # layer INPUT tokens (decoder.py:211-214/:304)
            new_mem.append(x_cat)
            mem_l = torch.cat((mem_vals[l], prepare_y(sd, b, x_cat, memory_mode, opq)), dim=1)
        else:
            mem_l = mem_vals[l]
        nxt = []
        for g, c in enumerate(cur):
            outs = []
            for j in range(nimgs[g]):
                if use_mask and causal:  # make_attn_mask decoder.py:389-433 (old memory labels are all < mem_nimgs <= idx)
                    assert len(cur) == 1
                    if Nm == 0 and j == 0:   # :399-402: idx of view 0 becomes 2 -> labels {1} (label 0 is its own)
                        y = mem_l[:, Ns[g]:2 * Ns[g]]
                    else:
                        y = mem_l[:, :Nm + offs[g][j]]
                elif use_mask:  # a view never cross-attends to its own new tokens (make_mem_mask decoder.py:119-139)
                    lo = Nm + offs[g][j]
                    y = torch.cat((mem_l[:, :lo], mem_l[:, lo + Ns[g]:]), dim=1)
                else:
                    y = mem_l
                outs.append(decoder_block(sd, b, c[j:j + 1], y, poss[g].reshape(nimgs[g], Ns[g], 2)[j:j + 1],
                                          cfg, memory_mode, opq, sdpa))
            nxt.append(torch.cat(outs, dim=0))
        cur = nxt
        for g in range(len(cur)):
            feats[g].append(cur[g].reshape(1, nimgs[g], Ns[g], D))
    if not render:
        # run_feedback_layers feedback_mechanism.py:39-53
        if "feedback_layer.fc1.weight" in sd:          # 'single_mlp' (feedback_mechanism.py:12-14)
            fb = layer_norm(new_mem[-1], sd["feedback_norm.weight"], sd["feedback_norm.bias"], 1e-5)
            offset = mlp(sd, "feedback_layer", fb, opq)
            new_mem = [m + offset for m in new_mem[:-1]] + [new_mem[-1]]
        elif "feedback_layer.weight" in sd:           # 'single_linear' (:15-17)
            fb = layer_norm(new_mem[-1], sd["feedback_norm.weight"], sd["feedback_norm.bias"], 1e-5)
            offset = linear(fb, sd["feedback_layer.weight"], sd["feedback_layer.bias"], opq)
            new_mem = [m + offset for m in new_mem[:-1]] + [new_mem[-1]]
        # else: no feedback layer, run_feedback_layers returns mem unchanged (:45-46)
        mem_out = [torch.cat((mem_vals[l], prepare_y(sd, f"blocks_dec.{l}", new_mem[l], memory_mode, opq)), dim=1)
                   for l in range(cfg.dec_depth)]  # decoder.py:236-239 / :327-330
        labels = []
        k = 0
        for n, N in zip(nimgs, Ns):  # decoder.py:241-249 / :332-334
            labels.append((torch.arange(n, dtype=torch.int64) + mem_nimgs + k).repeat_interleave(N).view(1, -1))
            k += n
        mem_labels = torch.cat([mem_labels] + labels, dim=1)
        tot = mem_nimgs + sum(nimgs)
        if causal:   # decoder.py:461-464: protected images / tokens are bookkeeping of the training-time dropout
            prot = min(protected_imgs, mem_prot_imgs + sum(nimgs))
            out_mem = (mem_out, mem_labels, tot, prot, mem_prot_tok + (prot - mem_prot_imgs) * Ns[0])
        else:
            out_mem = (mem_out, mem_labels, tot, tot, mem_labels.shape[1])
    else:
        out_mem = (mem_vals, mem_labels, mem_nimgs, mem_prot_imgs, mem_prot_tok)  # decoder.py:252 / :339
    pms = []
    for g in range(len(cur)):
        Hh, Ww = (int(v) for v in shapes[g].reshape(-1, 2)[0])
        pms.append(unpatchify_head(sd, cfg, cur[g], Hh, Ww).unsqueeze(0))
    if return_feats:
        # _compute_prediction_head normalises the last entry IN the list (decoder.py:150); views as decoder.py:346 / :260
        for g in range(len(cur)):
            feats[g][-1] = layer_norm(cur[g], sd["norm_dec.weight"], sd["norm_dec.bias"], 1e-6).reshape(1, nimgs[g], Ns[g], D)
        return out_mem, (pms if is_list else pms[0]), (feats if is_list else feats[0])
    return out_mem, (pms if is_list else pms[0])


def postprocess(pointmaps):
    """engine/inference.py:16-27 (activation part) + tools/geometry.py:14-18 apply_exp_to_norm."""
    pm = pointmaps.float()

    def norm_exp(v):
        d = v.norm(dim=-1, keepdim=True)
        return v / d.clip(min=1e-8) * torch.expm1(d)
    return {"pts3d": norm_exp(pm[..., 0:3]), "pts3d_local": norm_exp(pm[..., 3:6]), "conf": 1.0 + pm[..., 6].exp()}


# ------------------------------------------------------------------------------------------------
# scene schedule (the BASELINE unit of work)
# ------------------------------------------------------------------------------------------------
def run_scene(sd_enc, sd_dec, cfg, imgs, true_shape, mem_batches=None, memory_mode="kv", opq=_ID, sdpa=True,
              timings=None):
    """One scene: encode V views, memory update with the demo schedule ``[2,1,...,1]``
    (demo/inference.py:188-191; loop engine/inference.py:396-442), render all V views against the final
    memory (engine/inference.py:489-522), fp32 activation (engine/inference.py:19-27).

    Single aspect ratio.  Returns (update_pointmaps [V,H,W,7], render_pointmaps [V,H,W,7], mem_tuple)."""
    import time
    V = imgs.shape[0]
    if mem_batches is None:
        mem_batches = [2] + [1] * (V - 2) if V >= 2 else [1]
    t0 = time.perf_counter()
    xs, poss = [], []
    for v in range(V):  # one view per encoder call, like the demo (max_bs=1)
        xv, pv = encoder_forward(sd_enc, cfg, imgs[v:v + 1], true_shape[v:v + 1], opq, sdpa)
        xs.append(xv)
        poss.append(pv)
    x = torch.cat(xs, dim=0)
    pos = torch.cat(poss, dim=0)
    t1 = time.perf_counter()
    mem = None
    upd = []
    i = 0
    for nb in mem_batches:
        mem, pm = decoder_forward(sd_dec, cfg, x[i:i + nb].unsqueeze(0), pos[i:i + nb].unsqueeze(0),
                                  true_shape[i:i + nb].unsqueeze(0), mem, False, memory_mode, opq, sdpa)
        upd.append(pm[0])
        i += nb
    t2 = time.perf_counter()
    ren = []
    for v in range(V):
        _, pm = decoder_forward(sd_dec, cfg, x[v:v + 1].unsqueeze(0), pos[v:v + 1].unsqueeze(0),
                                true_shape[v:v + 1].unsqueeze(0), mem, True, memory_mode, opq, sdpa)
        ren.append(pm[0])
    t3 = time.perf_counter()
    if timings is not None:
        timings.update(encode=t1 - t0, update=t2 - t1, render=t3 - t2)
    return torch.cat(upd, dim=0), torch.cat(ren, dim=0), mem
