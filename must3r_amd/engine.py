"""Callers' side of the forward path, restated MI355X-first.

* ``postprocess`` -- must3r/engine/inference.py:16-48, activation part (:19-27) as one fused device kernel.
* ``run_scene``   -- the BASELINE unit of work: encode V views, update the memory with
  the demo schedule ``[2,1,...,1]`` (demo/inference.py:188-191; loop engine/inference.py:396-442), render all V
  views against the final memory (engine/inference.py:489-522), fp32 activation.  Views are independent in the
  encoder and in the render pass, so both run as ONE batched native call instead of the reference's per-view
  Python loop; the memory update is inherently sequential and stays a loop of decoder calls.
"""
import torch

from . import _lib
from .model import ActivationType


@torch.no_grad()
def postprocess(pointmaps, pointmaps_activation=ActivationType.NORM_EXP, compute_cam=False):
    """pointmaps [...,H,W,7] fp32 (cuda) -> dict(pts3d [...,3], pts3d_local [...,3], conf [...]) and, with
    ``compute_cam`` (engine/inference.py:29-47), ``focal`` [...] and ``c2w`` [...,4,4]: Weiszfeld focal of
    ``pts3d_local`` about the principal point (W/2, H/2) and the conf-1 weighted rigid registration
    ``pts3d_local -> pts3d``, all in the same native call as the activation (``must3r_hip_postprocess_cam``)."""
    if isinstance(pointmaps_activation, str):
        pointmaps_activation = ActivationType(pointmaps_activation)
    if pointmaps_activation not in (ActivationType.NORM_EXP, ActivationType.LINEAR):
        raise ValueError(f"Unknown activation: {pointmaps_activation}")      # head.py:21
    if pointmaps.shape[-1] != 7:
        raise NotImplementedError("fused postprocess handles the 7-channel (pts3d, pts3d_local, conf) layout of the released models")
    act = _lib.ACT_LINEAR if pointmaps_activation == ActivationType.LINEAR else _lib.ACT_NORM_EXP
    if not pointmaps.is_cuda:
        raise RuntimeError("must3r_amd.postprocess: input is on CPU; the HIP path has no CPU fallback")
    pm = pointmaps.float().contiguous()
    lead = pm.shape[:-1]
    npix = pm.numel() // 7
    p3 = torch.empty((*lead, 3), dtype=torch.float32, device=pm.device)
    pl = torch.empty((*lead, 3), dtype=torch.float32, device=pm.device)
    cf = torch.empty(lead, dtype=torch.float32, device=pm.device)
    lib = _lib.load()
    stream = torch.cuda.current_stream(pm.device).cuda_stream
    out = {"pts3d": p3, "pts3d_local": pl, "conf": cf}
    if not compute_cam:
        with torch.cuda.device(pm.device):   # launch from the tensor's device whatever the caller's current device is
            _lib.check(lib.must3r_hip_postprocess_act(pm.data_ptr(), act, p3.data_ptr(), pl.data_ptr(), cf.data_ptr(), npix, stream))
        return out
    if pm.dim() < 3:
        raise ValueError("compute_cam needs pointmaps of shape [..., H, W, 7]")
    batch_dims = tuple(pm.shape[:-3])
    H, W = int(pm.shape[-3]), int(pm.shape[-2])
    n = npix // (H * W) if H * W else 0
    focal = torch.empty(batch_dims, dtype=torch.float32, device=pm.device)
    c2w = torch.empty((*batch_dims, 4, 4), dtype=torch.float32, device=pm.device)
    if n:
        with torch.cuda.device(pm.device):
            nbytes = lib.must3r_hip_postprocess_cam_scratch_bytes(n, H, W)
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=pm.device)
            _lib.check(lib.must3r_hip_postprocess_cam_act(pm.data_ptr(), act, n, H, W, p3.data_ptr(), pl.data_ptr(), cf.data_ptr(),
                                                      focal.data_ptr(), c2w.data_ptr(), scratch.data_ptr(), nbytes, stream))
    out["focal"] = focal
    out["c2w"] = c2w
    return out


def demo_mem_batches(n_views, init_num_images=2, batch_num_views=1):
    """demo/inference.py:188-191."""
    if n_views <= init_num_images:
        return [n_views]
    rest = n_views - init_num_images
    out = [init_num_images] + [batch_num_views] * (rest // batch_num_views)
    if rest % batch_num_views:
        out.append(rest % batch_num_views)
    return out


@torch.no_grad()
def run_scene(encoder, decoder, imgs, true_shape, mem_batches=None, render_bs=None, activate=True, encoder_tokens=None):
    """One scene with a single aspect ratio.  imgs fp32 [V,3,H,W] (cuda), true_shape int64 [V,2].

    (Rounds 1-2 carried an ``overlap`` option -- the encoder of the later views on a second stream under the memory update of the
    earlier ones: +3 % with the 4-wave GEMMs, nothing with the 8-wave one-block-per-CU GEMM; removed in r03)

    Returns dict(update=[V,H,W,7], render=[V,H,W,7], mem=mem_tuple, x, pos[, pts3d, pts3d_local, conf of the render])."""
    V = imgs.shape[0]
    if mem_batches is None:
        mem_batches = demo_mem_batches(V)
    # The decoder needs (H, W) as host integers.  Reading them from a CUDA tensor is a device->host sync per call, which
    # stops the host from queueing ahead of the GPU (the reference does the same: head.py:33 `.cpu().tolist()`); one copy
    # per scene instead.
    true_shape = true_shape.cpu() if true_shape.is_cuda else true_shape
    x, pos = encoder_tokens if encoder_tokens is not None else encoder(imgs, true_shape)
    mem = None
    upd = []
    i = 0
    if imgs.is_cuda and hasattr(decoder, "reserve_memory_tokens"):
        decoder.reserve_memory_tokens = sum(mem_batches) * ((imgs.shape[-2] // 16) * (imgs.shape[-1] // 16))   # final memory size
    for nb in mem_batches:
        mem, pm = decoder(x[i:i + nb].unsqueeze(0), pos[i:i + nb].unsqueeze(0), true_shape[i:i + nb].unsqueeze(0), mem)
        upd.append(pm[0])
        i += nb
    ren = []
    bs = render_bs or V
    for v0 in range(0, V, bs):
        _, pm = decoder(x[v0:v0 + bs].unsqueeze(0), pos[v0:v0 + bs].unsqueeze(0), true_shape[v0:v0 + bs].unsqueeze(0), mem,
                        render=True)
        ren.append(pm[0])
    out = {"update": torch.cat(upd, dim=0), "render": torch.cat(ren, dim=0), "mem": mem, "x": x, "pos": pos}
    if activate:
        out.update(postprocess(out["render"]))
    return out


@torch.no_grad()
def run_scenes(encoder, decoder, imgs, true_shape, mem_batches=None, activate=True, encoder_tokens=None):
    """S independent scenes of the same shape IN FLIGHT TOGETHER: ``imgs`` fp32 [S,V,3,H,W] (cuda), ``true_shape`` int64 [V,2].

    The schedule of every scene is ``run_scene``'s; the scenes ride the batch dimension of the reference's decoder API
    (decoder.py:170-186: ``x[i]`` is ``[B, nimg, Ni, Denc]``; the scenes of a batch never interact), which the native decoder runs
    as ONE launch sequence per call: the strictly sequential memory update -- one view per call, M = 768 rows per GEMM for a
    lone scene -- becomes M = S x 768, each weight tile staged once per S x the rows.  Encoder and render are batched over all
    S x V views like ``run_scene`` batches V.

    Returns dict(update=[S,V,H,W,7], render=[S,V,H,W,7], mem=mem_tuple with [S, Nm, mem_D] tensors, x, pos[, pts3d, ...])."""
    S, V = int(imgs.shape[0]), int(imgs.shape[1])
    if mem_batches is None:
        mem_batches = demo_mem_batches(V)
    true_shape = true_shape.cpu() if true_shape.is_cuda else true_shape
    if encoder_tokens is not None:
        x, pos = encoder_tokens
    else:
        x, pos = encoder(imgs.reshape(S * V, *imgs.shape[2:]), true_shape.repeat(S, 1))
    x = x.view(S, V, *x.shape[1:])
    pos = pos.view(S, V, *pos.shape[1:])
    ts = true_shape.unsqueeze(0).expand(S, -1, -1)
    if hasattr(decoder, "reserve_memory_tokens"):
        decoder.reserve_memory_tokens = sum(mem_batches) * ((imgs.shape[-2] // 16) * (imgs.shape[-1] // 16))
    # r05: the update pointmaps of all schedule steps land in ONE [S, V, H, W, 7] buffer -- every call gets its slice [:, i:i+nb] (batch stride V views) as
    # `pointmaps_out` -- instead of being concatenated afterwards (torch.cat copied 2.2 GB per 20-scene step, ~3 ms)
    strided = bool(getattr(decoder, "_ctx", "x") != "x") and imgs.is_cuda   # the native module takes `pointmaps_out`; stand-ins (the CPU oracle in tests) do not
    H, W = int(imgs.shape[-2]), int(imgs.shape[-1])
    upd_all = torch.empty((S, sum(mem_batches), H, W, 7), dtype=torch.float32, device=imgs.device) if strided else None
    mem, upd, i = None, [], 0
    for nb in mem_batches:
        if strided:
            mem, pm = decoder(x[:, i:i + nb], pos[:, i:i + nb], ts[:, i:i + nb], mem, pointmaps_out=upd_all[:, i:i + nb])
        else:
            mem, pm = decoder(x[:, i:i + nb], pos[:, i:i + nb], ts[:, i:i + nb], mem)
            upd.append(pm)
        i += nb
    _, ren = decoder(x, pos, ts, mem, render=True)
    out = {"update": upd_all if strided else torch.cat(upd, dim=1), "render": ren, "mem": mem, "x": x, "pos": pos}
    if activate:
        out.update(postprocess(out["render"]))
    return out


@torch.no_grad()
def run_scene_mixed(encoder, decoder, img_groups, mem_batches=None, activate=True):
    """One scene whose views come in several aspect ratios (BASELINE.json configs[4]: 512 x {384,336,288,256,160}).

    ``img_groups``: list of fp32 [n_g,3,H_g,W_g] cuda tensors, one per aspect ratio; the scene's view order is group order,
    then order inside the group.  Every group is encoded with one batched encoder call; the memory update walks the views
    with the demo schedule ``[2,1,...,1]`` (a batch that spans two aspect ratios becomes a ``forward_list`` call,
    decoder.py:158-265, like ``inference_multi_ar`` does it, engine/inference.py:396-442); the render pass is ONE
    ``forward_list`` call over all groups against the final memory (engine/inference.py:489-522).

    Returns dict(update=[per-view [H,W,7]], render=[per-group [n_g,H_g,W_g,7]], mem[, pts3d/pts3d_local/conf per group])."""
    dev = img_groups[0].device
    enc = []
    for g in img_groups:
        n, _, H, W = g.shape
        ts = torch.tensor([[H, W]] * n, dtype=torch.int64)
        x, pos = encoder(g, ts)
        enc.append((x, pos, ts))
    owner = [(gi, j) for gi, g in enumerate(img_groups) for j in range(g.shape[0])]
    V = len(owner)
    if mem_batches is None:
        mem_batches = demo_mem_batches(V)
    if hasattr(decoder, "reserve_memory_tokens"):
        decoder.reserve_memory_tokens = sum(int(enc[gi][0].shape[1]) for gi, _ in owner[:sum(mem_batches)])   # final memory size
    mem, upd, i = None, [], 0
    for nb in mem_batches:
        batch = owner[i:i + nb]
        gids = sorted({gi for gi, _ in batch})
        xs, ps, tss = [], [], []
        for gi in gids:
            js = [j for g2, j in batch if g2 == gi]
            x, pos, ts = enc[gi]
            xs.append(x[js[0]:js[-1] + 1].unsqueeze(0))
            ps.append(pos[js[0]:js[-1] + 1].unsqueeze(0))
            tss.append(ts[js[0]:js[-1] + 1].unsqueeze(0))
        if len(gids) == 1:
            mem, pm = decoder(xs[0], ps[0], tss[0], mem)
            pms = [pm]
        else:
            mem, pms = decoder(xs, ps, tss, mem)
        for pm in pms:
            upd += [pm[0, k] for k in range(pm.shape[1])]
        i += nb
    _, ren = decoder([e[0].unsqueeze(0) for e in enc], [e[1].unsqueeze(0) for e in enc], [e[2].unsqueeze(0) for e in enc], mem,
                     render=True)
    out = {"update": upd, "render": [r[0] for r in ren], "mem": mem}
    if activate:
        pp = [postprocess(r) for r in out["render"]]
        for k in ("pts3d", "pts3d_local", "conf"):
            out[k] = [p[k] for p in pp]
    return out


@torch.no_grad()
def run_scenes_mixed(encoder, decoder, img_groups, mem_batches=None, activate=True):
    """S independent mixed-resolution scenes IN FLIGHT TOGETHER (BASELINE.json configs[4] with every launch chip-filling): ``img_groups``
    is a list of fp32 [S, n_g, 3, H_g, W_g] cuda tensors, one per aspect ratio; scene b is ``[g[b] for g in img_groups]`` with
    ``run_scene_mixed``'s view order and schedule.  The scenes ride the batch dimension of ``forward_list`` (decoder.py:158-265 takes
    lists of [B, nimg_i, N_i, C]); encoder calls are batched over S x n_g views per aspect ratio.

    Returns dict(update=[per-view [S,H,W,7]], render=[per-group [S,n_g,H_g,W_g,7]], mem[, pts3d/pts3d_local/conf per group])."""
    S = int(img_groups[0].shape[0])
    enc = []
    for g in img_groups:
        _, n, _, H, W = g.shape
        ts = torch.tensor([[H, W]] * n, dtype=torch.int64)
        x, pos = encoder(g.reshape(S * n, 3, H, W), ts.repeat(S, 1))
        enc.append((x.view(S, n, *x.shape[1:]), pos.view(S, n, *pos.shape[1:]), ts.unsqueeze(0).expand(S, -1, -1)))
    owner = [(gi, j) for gi, g in enumerate(img_groups) for j in range(g.shape[1])]
    V = len(owner)
    if mem_batches is None:
        mem_batches = demo_mem_batches(V)
    if hasattr(decoder, "reserve_memory_tokens"):
        decoder.reserve_memory_tokens = sum(int(enc[gi][0].shape[2]) for gi, _ in owner[:sum(mem_batches)])
    mem, upd, i = None, [], 0
    for nb in mem_batches:
        batch = owner[i:i + nb]
        gids = sorted({gi for gi, _ in batch})
        xs, ps, tss = [], [], []
        for gi in gids:
            js = [j for g2, j in batch if g2 == gi]
            x, pos, ts = enc[gi]
            xs.append(x[:, js[0]:js[-1] + 1].contiguous())
            ps.append(pos[:, js[0]:js[-1] + 1].contiguous())
            tss.append(ts[:, js[0]:js[-1] + 1])
        if len(gids) == 1:
            mem, pm = decoder(xs[0], ps[0], tss[0], mem)
            pms = [pm]
        else:
            mem, pms = decoder(xs, ps, tss, mem)
        for pm in pms:
            upd += [pm[:, k] for k in range(pm.shape[1])]
        i += nb
    _, ren = decoder([e[0] for e in enc], [e[1] for e in enc], [e[2] for e in enc], mem, render=True)
    out = {"update": upd, "render": list(ren), "mem": mem}
    if activate:
        pp = [postprocess(r) for r in out["render"]]
        for k in ("pts3d", "pts3d_local", "conf"):
            out[k] = [p[k] for p in pp]
    return out


# ------------------------------------------------------------------------------------------------------------------
# memory surgery of the L3 engine -- same semantics as the reference helpers
# engine/inference.py:205-228, but the per-layer buffers are compacted IN PLACE, so the tensors stay prefix views of
# the decoder's over-allocated K|V buffers and the next memory update appends without copying the whole memory.
# ------------------------------------------------------------------------------------------------------------------
def _label_runs(mem_labels):
    """Host mirror of the label layout, ``[(label, n_tokens), ...]`` in row order, attached by the decoder to the label
    tensors it returns (``MUSt3R._forward_scene``) and kept up to date by the helpers below.  None when the caller built
    the labels itself: the helpers then fall back to the reference's boolean-mask indexing (a device->host sync each)."""
    runs = getattr(mem_labels, "_m3r_runs", None)
    if runs is None or mem_labels.dim() != 2 or mem_labels.shape[0] != 1 or sum(c for _, c in runs) != mem_labels.shape[1]:
        return None
    # an in-place edit of the labels outside these helpers (the reference's own _restore_label_in_mem does that, user code may) leaves
    # the attribute on the tensor but makes it stale: the mirror is only trusted while the tensor's version counter is the one it was
    # attached at
    if getattr(mem_labels, "_m3r_runs_version", None) != mem_labels._version:
        return None
    return runs


def attach_label_runs(mem_labels, runs):
    """Attach the host mirror of the label layout to ``mem_labels`` (valid until the tensor is next modified in place)."""
    mem_labels._m3r_runs = runs
    mem_labels._m3r_runs_version = mem_labels._version
    return mem_labels


def _ranges_of(runs, idx):
    out, off = [], 0
    for lab, cnt in runs:
        if lab == idx:
            out.append((off, off + cnt))
        off += cnt
    return out


def remove_from_mem(mem_values, mem_labels, idx):
    """``_remove_from_mem`` (engine/inference.py:205-213): drop every token whose label equals ``idx``.

    mem_values: list of [B=1, Nm, D]; mem_labels: int64 [1, Nm].  Returns (mem_values, mem_labels).

    When the tensors are the newest prefix views of the decoder's own buffers they are compacted IN PLACE (rows behind the
    first dropped one move up; the shorter views alias the same storage, so OLDER, longer views of these buffers change
    under the caller -- the L3 drivers never keep them).  With the host mirror of the labels the kept-row index is built
    on the host and uploaded: no ``nonzero`` read-back, the host keeps queueing ahead of the GPU."""
    B, Nm, D = mem_values[0].shape
    owner = getattr(mem_values[0], "_m3r_owner", None)
    in_place = (B == 1 and owner is not None and owner.valid == Nm
                and all(getattr(v, "_m3r_owner", None) is owner for v in mem_values))
    runs = _label_runs(mem_labels)
    if runs is not None:
        drop = _ranges_of(runs, idx)
        new_runs = [(lab, cnt) for lab, cnt in runs if lab != idx]
        if not drop:
            return mem_values, mem_labels
        first = drop[0][0]
        tail, off = [], 0
        for lab, cnt in runs:
            if lab != idx and off >= first:
                tail.append(torch.arange(off, off + cnt))
            off += cnt
        n = sum(c for _, c in new_runs)
        dev = mem_labels.device
        tail_idx = (torch.cat(tail) if tail else torch.zeros(0, dtype=torch.int64)).to(dev, non_blocking=True)
        if in_place:
            if n > first:
                for b in owner.bufs:
                    b[:, first:n] = b[:, tail_idx]     # gather into a temporary, then prefix write: sources are read first
            owner.valid = n
            vals = owner.views(n)
        else:
            vals = [torch.cat([v[:, :first], v[:, tail_idx]], dim=1) for v in mem_values]
        labels = torch.cat([mem_labels[:, :first], mem_labels[:, tail_idx]], dim=1)
        attach_label_runs(labels, new_runs)
        return vals, labels
    keep = mem_labels != idx
    if not in_place:
        return [v[keep].view(B, -1, D) for v in mem_values], mem_labels[keep].view(B, -1)
    kept = keep[0].nonzero().flatten()
    n = int(kept.numel())
    for b in owner.bufs:
        b[:, :n] = b[:, kept]          # gather then prefix write: source rows are read before being overwritten
    owner.valid = n
    return owner.views(n), mem_labels[keep].view(1, -1)


def restore_label_in_mem(mem_labels, old_idx_to_restore, new_idx_to_remove):
    """``_restore_label_in_mem`` (engine/inference.py:216-219), in place."""
    runs = _label_runs(mem_labels)
    mem_labels[mem_labels == new_idx_to_remove] = old_idx_to_restore
    if runs is not None:
        attach_label_runs(mem_labels, [(old_idx_to_restore if lab == new_idx_to_remove else lab, cnt) for lab, cnt in runs])
    return mem_labels


def update_in_mem(old_values, new_values, old_labels, new_labels, old_idx, new_idx):
    """``_update_in_mem`` (engine/inference.py:222-228): overwrite the tokens labelled ``old_idx`` in ``old_values`` with
    the tokens labelled ``new_idx`` of ``new_values`` (in place; views of the decoder's buffers stay valid).  Slice copies
    when both label layouts are known on the host, the reference's boolean masks otherwise."""
    ro, rn = _label_runs(old_labels), _label_runs(new_labels)
    if ro is not None and rn is not None:
        dst, src = _ranges_of(ro, old_idx), _ranges_of(rn, new_idx)
        if len(dst) == len(src) and all(d[1] - d[0] == s_[1] - s_[0] for d, s_ in zip(dst, src)):
            for k in range(len(old_values)):
                for (d0, d1), (s0, s1) in zip(dst, src):
                    old_values[k][:, d0:d1] = new_values[k][:, s0:s1]
            return old_values
    old_mask = old_labels == old_idx
    new_mask = new_labels == new_idx
    for k in range(len(old_values)):
        old_values[k][old_mask] = new_values[k][new_mask]
    return old_values


def rewind_mem(mem_values):
    """Declare the (shorter) views ``mem_values`` the newest ones of their buffers again.

    A refinement pass (engine/inference.py:427-440) decodes a batch against the memory, copies the appended tokens over the
    batch's old ones and throws the appended copy away: the rows the decoder appended are scratch.  Without this the
    buffers would still count them and the next update would take the copy-out path.  Only call it when nothing keeps the
    longer views.  No-op for tensors that are not prefix views of the decoder's buffers."""
    owner = getattr(mem_values[0], "_m3r_owner", None)
    if owner is not None and len(mem_values) == len(owner.bufs) and all(
            getattr(v, "_m3r_owner", None) is owner and v.data_ptr() == b.data_ptr() for v, b in zip(mem_values, owner.bufs)):
        owner.valid = int(mem_values[0].shape[1])
    return mem_values


@torch.no_grad()
def run_video(encoder, decoder, imgs, true_shape, local_context_size=25, is_keyframe=lambda i: i % 3 == 0,
              init_num_images=2, encoder_tokens=None):
    """Online / streaming memory (BASELINE.json configs[3]), single pass, single aspect ratio: the schedule of
    ``inference_video_multi_ar`` (engine/inference.py:232-366 with num_refinements_iterations=0).

    Every frame updates the memory ([init_num_images, 1, 1, ...]); the frames of the first batch and every frame for
    which ``is_keyframe(id)`` holds (reference default: every 3rd, :236) stay; other frames are evicted once they leave
    the window of the last ``local_context_size`` frames (:336-339) and at the end of the pass (:356-361).  Eviction uses
    ``remove_from_mem`` (in-place compaction of the decoder's buffers when the memory is theirs).

    Returns (mem_tuple, pointmaps_0 [V,H,W,7], keyframe ids)."""
    from collections import deque
    ts_host = true_shape.cpu() if true_shape.is_cuda else true_shape
    x, pos = encoder_tokens if encoder_tokens is not None else encoder(imgs, true_shape)
    V = x.shape[0]
    mem = None
    img_labels, keyframes, working = {}, set(), deque()
    pointmaps_0 = []
    i = 0
    for nb in demo_mem_batches(V, init_num_images, 1):
        ids = list(range(i, i + nb))
        n_before = 0 if mem is None else mem[2]
        mem, pm = decoder(x[i:i + nb].unsqueeze(0), pos[i:i + nb].unsqueeze(0), ts_host[i:i + nb].unsqueeze(0), mem)
        pointmaps_0.append(pm[0])
        mem = list(mem)
        new_labels = [n_before + j for j in range(nb)]      # decoder.py:332-334: labels = arange(n) + mem_nimgs
        first = len(img_labels) == 0
        for j, vid in enumerate(ids):
            img_labels[vid] = new_labels[j]
            working.append(vid)
            if first or is_keyframe(vid):
                keyframes.add(vid)
        while len(working) > local_context_size:              # :336-339
            old = working.popleft()
            if old not in keyframes:
                mem[0], mem[1] = remove_from_mem(mem[0], mem[1], img_labels[old])
        mem[2] = len(img_labels)                              # :342 restore mem_nimgs
        mem = tuple(mem)
        i += nb
    mem = list(mem)
    while working:                                            # :356-361
        old = working.popleft()
        if old not in keyframes:
            mem[0], mem[1] = remove_from_mem(mem[0], mem[1], img_labels[old])
    return tuple(mem), torch.cat(pointmaps_0, dim=0), sorted(keyframes)
