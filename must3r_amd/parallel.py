"""Multi-GPU forward path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

What shards and what does not:
  * encoder      -- views are independent                       -> each rank encodes its own views
  * memory update-- strictly sequential in the keyframes        -> REPLICATED on every rank (deterministic, so all
                    ranks build bit-identical memories; cheaper than broadcasting 12 x [Nm,1536] K|V tensors)
  * render       -- each view reads the frozen memory           -> each rank renders its own views
The single exchange step is an **all-gather of the encoded keyframe tokens** (what every rank needs to run the
update): per keyframe N x enc_dim 16-bit = 1.5 MiB at 512x384.  xGMI is point-to-point and the payload is a few
MiB per rank, so one all-gather (direct peer writes) is the right collective -- no ring all-reduce anywhere.

Everything here is backend-agnostic (gloo on CPU in tests, RCCL on the GPU node).
"""
import torch
import torch.distributed as dist

from .engine import demo_mem_batches


def _world(group):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_range(n_items, rank, world):
    """Contiguous balanced shard [lo, hi) of ``n_items`` for ``rank``."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_counts(k, device, group=None):
    """Row counts of every rank (host ints), one tiny all-gather + one read-back."""
    rank, world = _world(group)
    if world == 1:
        return [int(k)]
    if device.type == "cuda" and dist.get_backend(group) == "gloo":
        device = torch.device("cpu")
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([int(k)], dtype=torch.int64, device=device), group=group)
    return [int(c) for c in torch.cat(counts).tolist()]


def all_gather_varlen(t, group=None, counts=None):
    """Concatenate ``t`` ([k_rank, ...], k may differ per rank, may be 0) over ranks, in rank order.  ``counts``: the row
    counts of all ranks when the caller already knows them (saves the count exchange and its device->host read-back)."""
    rank, world = _world(group)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return t      # no process group at all; an initialised group of ONE rank still runs the collective (a 1-GPU box exercises RCCL)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no CUDA all_gather: stage through the host (only used when ranks share a GPU in tests)
        return all_gather_varlen(t.cpu(), group, counts).to(t.device)
    if counts is None:
        counts = gather_counts(t.shape[0], t.device, group)
    assert counts[rank] == t.shape[0], (counts, rank, t.shape)
    kmax = max(counts)
    if kmax == 0:
        return t
    if min(counts) == 0:
        # a rank without rows may not know the trailing shape (e.g. the token width of an encoder it never ran): adopt the
        # shape of the ranks that have rows (one tiny exchange, only in this ragged case)
        dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else t.device
        mine = torch.zeros(8, dtype=torch.int64, device=dev)
        if t.shape[0] > 0:
            mine[0] = t.dim() - 1
            mine[1:t.dim()] = torch.tensor(tuple(t.shape[1:]), dtype=torch.int64, device=dev)
        shapes = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(shapes, mine, group=group)
        ref = next(sh for sh, c in zip(shapes, counts) if c > 0).tolist()
        trail = tuple(int(v) for v in ref[1:1 + int(ref[0])])
        if t.shape[0] == 0 and tuple(t.shape[1:]) != trail:
            t = t.new_zeros((0,) + trail)
    if min(counts) == kmax:
        pad = t.contiguous()                      # equal shards: no staging copy
    else:
        pad = torch.zeros((kmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
    # ONE flat destination (all_gather_into_tensor: a single collective writing every rank's block at its offset, no per-rank
    # tensor list and no concatenation afterwards); ragged shards are compacted with one index_select
    out = torch.empty((world * kmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    if min(counts) == kmax:
        return out
    keep = torch.cat([torch.arange(r * kmax, r * kmax + c) for r, c in enumerate(counts)]).to(t.device)
    return out.index_select(0, keep)


def _encode_local(encoder, imgs_local, true_shape_local):
    """Encode this rank's views; a rank without views (n_views < world) skips the call and returns empty [0,N,C] tokens."""
    if imgs_local.shape[0] > 0:
        return encoder(imgs_local, true_shape_local)
    N = (imgs_local.shape[-2] // 16) * (imgs_local.shape[-1] // 16)
    C = int(getattr(encoder, "embed_dim", 1024))
    return (torch.zeros((0, N, C), dtype=torch.float32, device=imgs_local.device),
            torch.zeros((0, N, 2), dtype=torch.int64, device=imgs_local.device))


def grid_positions(gh, gw, device):
    """croco PositionGetter: int64 [gh*gw, 2] = (y, x) of a row-major gh x gw token grid -- what the encoder
    returns as ``pos`` for one view, so positions never have to travel between ranks."""
    y = torch.arange(gh, dtype=torch.int64, device=device)
    x = torch.arange(gw, dtype=torch.int64, device=device)
    return torch.cartesian_prod(y, x)


def _gather_keyframes(x, pos, true_shape_local, keyframe_local, group, comm_dtype, counts=None, grid=None, many_ar=False):
    """ONE all-gather of the encoded keyframe tokens in rank order -- the scene's only exchange.  The keyframe selection is a host
    index list (no boolean-mask indexing on the device = no nonzero() sync).  ``counts``: keyframes per rank when the caller
    knows the schedule (a benchmark, a fixed keyframe stride): no count exchange at all.

    r04 ("pos is recomputable from the grid"): positions and true shapes no longer ride two more collectives.
    Each keyframe's payload is its [N, C] tokens plus ONE extra row whose first four entries are (h // 256, h % 256, w // 256, w % 256)
    of its true shape -- integers below 256, exact in bf16 and fp16 -- and the positions are rebuilt from the token grid
    (``grid`` = the encoder's patch size; r05: each keyframe's grid is (h / patch) x (w / patch) of ITS OWN gathered true shape)."""
    idx = torch.nonzero(keyframe_local.cpu()).flatten()
    if counts is None:
        counts = gather_counts(int(idx.numel()), x.device, group)
    else:
        counts = [int(c) for c in counts]
        assert counts[_world(group)[0]] == int(idx.numel()), (counts, int(idx.numel()))
    idx_d = idx.to(x.device)
    kx = x.index_select(0, idx_d)
    ts = true_shape_local.cpu().index_select(0, idx).to(torch.int64)
    if ts.numel() and (int(ts.max()) >= 65536 or int(ts.min()) < 0):   # two base-256 digits per extent ride in one 16-bit token row (exact below 65536)
        raise ValueError(f"true_shape {tuple(ts.max(dim=0).values.tolist())} does not fit the keyframe payload's (h // 256, h % 256, w // 256, w % 256) row")
    meta = torch.zeros((kx.shape[0], 1, kx.shape[2]), dtype=torch.float32)
    meta[:, 0, 0], meta[:, 0, 1] = ts[:, 0] // 256, ts[:, 0] % 256
    meta[:, 0, 2], meta[:, 0, 3] = ts[:, 1] // 256, ts[:, 1] % 256
    pay = torch.cat([kx, meta.to(kx.device)], dim=1)
    if comm_dtype is not None:
        pay = pay.to(comm_dtype)  # 16-bit on the wire; the decoder rounds its operands to this type anyway
    pay = all_gather_varlen(pay, group, counts)
    kx = pay[:, :-1].float()
    m = pay[:, -1, :4].float().round().to(torch.int64).cpu()            # host copy of the shapes (the decoder needs host integers)
    kts = torch.stack([m[:, 0] * 256 + m[:, 1], m[:, 2] * 256 + m[:, 3]], dim=1)
    if grid is None:   # single-process callers without images: fall back to the positions themselves
        return kx, all_gather_varlen(pos.index_select(0, idx_d), group, counts), kts
    # r05: every keyframe's positions come from ITS OWN true shape, which arrived with its tokens -- (h / p) x (w / p) row-major
    # grid, exactly what both patch embeds of the reference hand out (PatchEmbedDust3R: the stored H x W IS the true shape; ManyAR: a portrait
    # view is transposed before the projection and gets the transposed grid).  r04 used the LOCAL rank's stored image
    # grid for all of them: wrong as soon as ranks hold different stored shapes with the same token count (384 x 512 on one, 512 x 384 on another).
    patch = int(grid)
    N = int(kx.shape[1])
    kpos = torch.empty((kx.shape[0], N, 2), dtype=torch.int64, device=kx.device)
    shapes = kts.tolist()
    for hw in sorted(set(map(tuple, shapes))):
        gh, gw = hw[0] // patch, hw[1] // patch
        if gh * gw != N or hw[0] % patch or hw[1] % patch:
            raise ValueError(f"gathered keyframe of true shape {hw} does not have the {N} tokens of this stack (patch {patch})")
        sel = torch.tensor([i for i, s_ in enumerate(shapes) if tuple(s_) == hw], dtype=torch.int64, device=kx.device)
        kpos.index_copy_(0, sel, grid_positions(gh, gw, kx.device).unsqueeze(0).expand(sel.numel(), -1, -1).contiguous())
    return kx, kpos, kts


def _grid_of(encoder, imgs_local):
    """The patch size positions are rebuilt with (the gathered true shapes do the rest); second value kept for callers of the r04 signature."""
    many_ar = getattr(getattr(encoder, "patch_embed", None), "kind", "PatchEmbedDust3R") == "ManyAR_PatchEmbed"
    return int(getattr(encoder, "patch_size", 16) or 16), many_ar


def _render_local(decoder, x, pos, true_shape_local, mem, imgs_local):
    if x.shape[0] > 0:
        _, pm = decoder(x.unsqueeze(0), pos.unsqueeze(0), true_shape_local.cpu().unsqueeze(0), mem, render=True)
        return pm[0]
    return torch.zeros((0, imgs_local.shape[-2], imgs_local.shape[-1], 7), dtype=torch.float32, device=x.device)


@torch.no_grad()
def run_scene_sharded(encoder, decoder, imgs_local, true_shape_local, keyframe_local, group=None, comm_dtype=None,
                      mem_batches=None, gather_outputs=False, keyframe_counts=None):
    """One scene whose views are sharded over the ranks of ``group``.

    imgs_local [v,3,H,W], true_shape_local [v,2]: this rank's views (global order = rank order, then local order; v may be
    0 when there are fewer views than ranks).  keyframe_local bool[v]: which local views enter the memory.  Returns
    dict(render=[v,H,W,7] local pointmaps, mem=memory tuple (identical on all ranks), n_keyframes) (+ render_all when
    gather_outputs).  With all views keyframes this is the STRONG-scaling form of the 20-view benchmark scene.
    ``keyframe_counts``: keyframes per rank, when known on the host (skips the count exchange).
    """
    x, pos = _encode_local(encoder, imgs_local, true_shape_local)
    grid, many_ar = _grid_of(encoder, imgs_local)
    kx, kpos, kts = _gather_keyframes(x, pos, true_shape_local, keyframe_local, group, comm_dtype, keyframe_counts, grid, many_ar)
    K = kx.shape[0]
    if K == 0:
        raise ValueError("run_scene_sharded: no keyframe on any rank")
    mem = None
    i = 0
    if hasattr(decoder, "reserve_memory_tokens"):
        decoder.reserve_memory_tokens = K * kx.shape[1]     # final memory size: no growth copies
    for nb in (mem_batches or demo_mem_batches(K)):
        mem, _ = decoder(kx[i:i + nb].unsqueeze(0), kpos[i:i + nb].unsqueeze(0), kts[i:i + nb].unsqueeze(0), mem)
        i += nb
    out = {"mem": mem, "n_keyframes": K}
    out["render"] = _render_local(decoder, x, pos, true_shape_local, mem, imgs_local)
    if gather_outputs:
        out["render_all"] = all_gather_varlen(out["render"], group)
    return out


class ContextParallel:
    """Exchange side of the context-parallel cross attention (SURVEY.md section 8f "later"; include/must3r_hip.h ``must3r_hip_cp``): the memory of ONE scene is
    sharded over the ranks of ``group``; in a one-view memory update every rank attends its own rows, leaves one fp32 partial per layer -- un-normalised O plus
    (m, l) per query row and head -- in its slot of ``slots`` and the library calls back here for the all-gather (``all_gather_into_tensor`` on the caller's
    stream: RCCL over xGMI on the GPU node, gloo with host staging when the ranks of a dry run share a GPU), then merges the world's partials.

    ``n_mem_total`` (set by the driver before every call): memory rows over all ranks."""

    def __init__(self, group=None, partial16=True):
        self.group = group
        self.partial16 = bool(partial16)   # the 16-bit partial format (O / l in the operand type): half the bytes of the fp32 one on the links
        self.rank, self.world = _world(group)
        self.live = dist.is_available() and dist.is_initialized()
        self.n_mem_total = 0
        self.slots = None
        self.exchanges = 0
        self.bytes_gathered = 0
        self._error = None
        self._struct = None
        from . import _lib
        self._cb = _lib.CpExchangeFn(self._exchange)     # (kept alive with the object: the library calls it during must3r_hip_decode)

    def native_args(self, ctx, rows, device):
        from . import _lib
        need = int((ctx.lib.must3r_hip_cp_slot_bytes16 if self.partial16 else ctx.lib.must3r_hip_cp_slot_bytes)(ctx.handle, int(rows)))
        if self.slots is None or self.slots.device != device or self.slots.shape[1] < need:
            self.slots = torch.empty((self.world, need), dtype=torch.uint8, device=device)
        import ctypes as C
        self._struct = _lib.Cp(self.world, self.rank, int(self.n_mem_total), 1 if self.partial16 else 0, self.slots.data_ptr(), int(self.slots.shape[1]), self._cb, None)
        return self._struct

    def _exchange(self, user, layer, slots_ptr, slot_bytes, n_slots, my_slot, stream):
        try:
            buf = self.slots
            assert int(slots_ptr) == buf.data_ptr() and int(slot_bytes) == buf.shape[1] and int(n_slots) == self.world and int(my_slot) == self.rank
            self.exchanges += 1
            self.bytes_gathered += int(slot_bytes) * int(n_slots)
            if not self.live:
                return 0                                   # no process group: a world of one, the slot is already where it belongs
            if buf.is_cuda and dist.get_backend(self.group) == "gloo":
                out = torch.empty((self.world, buf.shape[1]), dtype=torch.uint8)      # ranks sharing a GPU (dry run / tests): through the host
                dist.all_gather_into_tensor(out.view(-1), buf[self.rank].cpu(), group=self.group)
                buf.copy_(out.to(buf.device))
            else:
                # in place: rank r's contribution already sits at slot r of the receive buffer (ncclAllGather's in-place form: sendbuff = recvbuff + rank * count)
                dist.all_gather_into_tensor(buf.view(-1), buf[self.rank], group=self.group)
            return 0
        except BaseException as e:   # noqa: BLE001 -- ctypes cannot propagate it: kept for reraise(), the native call fails with status 1
            self._error = e
            return 1

    def reraise(self):
        e, self._error = self._error, None
        if e is not None:
            raise e


def _video_context_parallel(decoder, ax, apos, ats, group, local_context_size, is_keyframe, init_num_images, partial16=True):
    """The streaming schedule of ``engine.run_video`` with the memory SHARDED over the ranks: image label L lives on rank L % world.  The init call (several views
    attending each other's pre-feedback tokens) is replicated and every rank keeps its own labels' rows; every later one-view call runs context-parallel
    (``decoder(.., cp=..)``): all ranks compute the frame, the owner of its label keeps the appended rows, the others rewind their buffers.  Eviction touches
    the owner only.  Returns (local memory tuple, pointmaps_0 (identical on all ranks), keyframe ids, rows held per rank)."""
    from collections import deque
    from .engine import demo_mem_batches, remove_from_mem
    rank, world = _world(group)
    cpx = ContextParallel(group, partial16=partial16)
    V, N = int(ax.shape[0]), int(ax.shape[1])
    ts_host = ats.cpu() if ats.is_cuda else ats
    mem = None
    img_labels, keyframes, working, live = {}, set(), deque(), {}
    pointmaps_0 = []
    i = 0
    for nb in demo_mem_batches(V, init_num_images, 1):
        ids = list(range(i, i + nb))
        n_before = len(img_labels)
        if mem is None:
            mem, pm = decoder(ax[i:i + nb].unsqueeze(0), apos[i:i + nb].unsqueeze(0), ts_host[i:i + nb].unsqueeze(0), None)
            mem = list(mem)
            for j in range(nb):
                live[n_before + j] = N
                if (n_before + j) % world != rank:
                    mem[0], mem[1] = remove_from_mem(mem[0], mem[1], n_before + j)
        else:
            assert nb == 1, "the context-parallel call takes one view"
            cpx.n_mem_total = sum(live.values())
            old_vals, old_labels = mem[0], mem[1]
            new_mem, pm = decoder(ax[i:i + 1].unsqueeze(0), apos[i:i + 1].unsqueeze(0), ts_host[i:i + 1].unsqueeze(0), tuple(mem), cp=cpx)
            live[n_before] = N
            if n_before % world == rank:
                mem = list(new_mem)
            else:   # not this rank's frame: the rows the call appended are scratch -- keep the (possibly re-allocated) buffers, declare the old length valid again
                n_old = int(old_labels.shape[1])
                own = getattr(new_mem[0][0], "_m3r_owner", None)
                if own is not None:
                    own.valid = n_old
                    vals = own.views(n_old)
                else:
                    vals = [v[:, :n_old] for v in new_mem[0]]
                mem = [vals, old_labels, mem[2], mem[3], mem[4]]
        pointmaps_0.append(pm[0])
        first = len(img_labels) == 0
        for j, vid in enumerate(ids):
            img_labels[vid] = n_before + j
            working.append(vid)
            if first or is_keyframe(vid):
                keyframes.add(vid)
        while len(working) > local_context_size:
            old = working.popleft()
            if old not in keyframes:
                live.pop(img_labels[old], None)
                mem[0], mem[1] = remove_from_mem(mem[0], mem[1], img_labels[old])     # a no-op on the ranks that do not hold the label
        mem[2] = mem[3] = len(img_labels)                                             # global image count: the next call's labels are arange(n) + mem[2]
        mem[4] = int(mem[1].shape[1])
        i += nb
    while working:
        old = working.popleft()
        if old not in keyframes:
            live.pop(img_labels[old], None)
            mem[0], mem[1] = remove_from_mem(mem[0], mem[1], img_labels[old])
    mem[4] = int(mem[1].shape[1])
    rows = [sum(n for lab, n in live.items() if lab % world == r) for r in range(world)]
    assert rows[rank] == int(mem[1].shape[1]), (rows, rank, tuple(mem[1].shape))
    return tuple(mem), torch.cat(pointmaps_0, dim=0), sorted(keyframes), rows, cpx


def gather_memory(mem, rows, group=None):
    """All ranks' shards of a sharded memory -> the whole memory on every rank (rank order), for the view-sharded render pass."""
    vals = [all_gather_varlen(v[0], group, rows).unsqueeze(0) for v in mem[0]]
    labels = all_gather_varlen(mem[1][0], group, rows).unsqueeze(0)
    return (vals, labels, mem[2], mem[3], int(labels.shape[1]))


@torch.no_grad()
def run_video_sharded(encoder, decoder, imgs_local, true_shape_local, group=None, comm_dtype=None, local_context_size=25,
                      is_keyframe=lambda i: i % 3 == 0, init_num_images=2, render=True, gather_outputs=False, frame_counts=None,
                      context_parallel=False):
    """Online / streaming memory over a sharded frame sequence (BASELINE.json configs[3]; schedule of
    ``inference_video_multi_ar``, engine/inference.py:232-366, via ``engine.run_video``).

    The frames of the sequence are dealt to the ranks in contiguous blocks (global frame id = rank order, then local order).
    Encoding -- the only per-frame work that does not depend on the memory -- is sharded; the encoded tokens of ALL frames
    are exchanged with one all-gather (every frame updates the memory in the online schedule, so every rank needs every
    frame's tokens); the sequential memory update with the sliding window / keyframe eviction is replicated
    (deterministic -> identical memories, nothing else to exchange); the final render of every frame against the final
    keyframe memory (engine/inference.py:489-522) is sharded again.  Returns dict(mem, keyframes, pointmaps_0 (update-pass
    pointmaps of ALL frames, identical on every rank), render (local frames)[, render_all]).

    ``context_parallel=True`` (r06): the memory itself is sharded over the ranks and the per-frame cross attention runs context-parallel
    (``_video_context_parallel``): per-rank cross-attention work and K|V memory footprint drop by the world size, for one all-gather of
    [tokens, dec_dim] 16-bit + [tokens, 2 heads] fp32 partials per layer and frame (``context_parallel="fp32"``: fp32 partials, twice the bytes)."""
    from .engine import run_video
    x, pos = _encode_local(encoder, imgs_local, true_shape_local)
    all_kf = torch.ones(x.shape[0], dtype=torch.bool)
    grid, many_ar = _grid_of(encoder, imgs_local)
    ax, apos, ats = _gather_keyframes(x, pos, true_shape_local, all_kf, group, comm_dtype, frame_counts, grid, many_ar)
    if context_parallel:
        # r06 (SURVEY.md section 8f "later"): the memory is SHARDED over the ranks (label L on rank L % world) and the per-frame update attends it context-parallel:
        # one all-gather of fp32 partials per layer and frame instead of a replicated cross attention over the whole memory; the whole memory is gathered once, at
        # the end, for the view-sharded render
        mem_local, pm0, keyframes, rows, cpx = _video_context_parallel(decoder, ax, apos, ats, group, local_context_size, is_keyframe, init_num_images,
                                                                       partial16=(context_parallel != "fp32"))
        mem = gather_memory(mem_local, rows, group) if (render or gather_outputs) else mem_local
        out = {"mem": mem, "mem_local": mem_local, "rows_per_rank": rows, "keyframes": keyframes, "pointmaps_0": pm0,
               "cp_exchanges": cpx.exchanges, "cp_bytes_gathered": cpx.bytes_gathered}
    else:
        mem, pm0, keyframes = run_video(None, decoder, None, ats, local_context_size=local_context_size, is_keyframe=is_keyframe,
                                        init_num_images=init_num_images, encoder_tokens=(ax, apos))
        out = {"mem": mem, "keyframes": keyframes, "pointmaps_0": pm0}
    if render:
        out["render"] = _render_local(decoder, x, pos, true_shape_local, mem, imgs_local)
        if gather_outputs:
            out["render_all"] = all_gather_varlen(out["render"], group)
    return out
