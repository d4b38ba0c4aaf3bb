"""The reference's L3 drivers of the forward path under their own names (SURVEY.md section 8a row a19,
must3r/engine/inference.py:51-688): the schedules of encoder / decoder calls that turn a list of views of any aspect ratio
into a memory and into pointmaps.  Host logic only -- every tensor operation is a call into the drop-in modules
(``must3r_amd.model``) or into ``must3r_amd.engine``'s native helpers -- restated so that the host never waits for the GPU
on the way:

* the aspect-ratio grouping (``stack_views``) is done once on host integers (the reference indexes the result of a
  device ``torch.unique`` element by element: one device->host sync per view);
* the labels a memory update gives its views are known on the host (decoder.py:241-249, :332-334: ``arange(n) +
  mem_nimgs``, group after group), so nothing is read back to find them (engine/inference.py:290, :426);
* memory surgery goes through ``engine.{remove_from_mem, restore_label_in_mem, update_in_mem}``: the decoder's K|V
  buffers are compacted / overwritten in place and stay appendable (no ``torch.concatenate`` of the whole memory);
* images are only moved to the device when the encoder actually needs them.

Same arguments, return values, label bookkeeping and quirks as the reference (label ``j`` of a batch goes to the
``j``-th image of the batch in *input* order even when the aspect-ratio grouping reordered the decoder's groups,
:297/:444; the reference image -- label 0 -- is never refreshed, :318/:432).  Checked call for call against the
reference's own drivers running on the same modules (tests/test_oracle_vs_reference.py).
"""
from collections import deque

import torch

from .engine import remove_from_mem, restore_label_in_mem, update_in_mem, rewind_mem


def get_Nmem(mem):
    """engine/inference.py:530-535."""
    return 0 if mem is None else int(mem[1].shape[1])


def _default_post_process(x):  # engine/inference.py:171
    return {'pts3d': x}


def _is_missing(v):
    return v is None


def _host(t):
    return t.cpu() if t.is_cuda else t


# ------------------------------------------------------------------------------------------------------------------
# grouping by aspect ratio
# ------------------------------------------------------------------------------------------------------------------
def stack_views(true_shape, values, max_bs=None):
    """engine/inference.py:65-136.  ``true_shape`` int [n,2]; ``values``: list of per-view sequences (tensors or None).

    Views are grouped by (H, W), groups in ascending lexicographic order (what ``torch.unique(dim=0)`` yields), views
    in input order inside a group.  In a group where only *some* views miss a value (None: encoder tokens still to be
    computed), those views move to a group of their own appended after the complete ones (:84-107).  With ``max_bs`` every
    group is cut into chunks of at most ``max_bs`` views (:116-127).  Returns ``(true_shape_stacks, index_stacks,
    *value_stacks)``: per group a stacked tensor -- or None where a value is missing (:129-134)."""
    host = _host(true_shape)   # the only device->host copy of the grouping
    keys = [tuple(row) for row in host.tolist()]
    by_shape = {}
    for i, k in enumerate(keys):
        by_shape.setdefault(k, []).append(i)
    groups = [by_shape[k] for k in sorted(by_shape)]

    def incomplete(i):
        return any(_is_missing(v[i]) for v in values)

    moved = []
    for g, members in enumerate(groups):
        lost = [i for i in members if incomplete(i)]
        if lost and len(lost) != len(members):
            groups[g] = [i for i in members if not incomplete(i)]
            moved.append(lost)
    groups += moved
    if max_bs is not None:
        groups = [members[a:a + max_bs] for members in groups for a in range(0, len(members), max_bs)]

    def pack(seq, members):
        items = [seq[i] for i in members]
        if any(_is_missing(e) for e in items):
            return None
        return torch.stack(items, dim=0)

    # shape stacks are host tensors: both modules read (H, W) as host integers (head.py:33; the native encoder / decoder)
    shape_stacks = [torch.stack([host[i] for i in members], dim=0) for members in groups]
    return (shape_stacks, groups, *[[pack(seq, members) for members in groups] for seq in values])


def unstack_pointmaps(index_stacks_i, pointmaps_0_i):
    """engine/inference.py:538-551: per-group dicts of [n_g, ...] tensors -> per-view dicts, in input order."""
    n = max(max(idx) for idx in index_stacks_i) + 1
    out = [None] * n
    for group, idx in zip(pointmaps_0_i, index_stacks_i):
        for j, i in enumerate(idx):
            out[i] = {k: v[j] for k, v in group.items()}
    return out


# ------------------------------------------------------------------------------------------------------------------
# one encoder pass / one decoder call over several aspect ratios
# ------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def encoder_multi_ar(encoder, imgs, true_shape, verbose=False, max_bs=None, device=None, preserve_gpu_mem=False):
    """engine/inference.py:139-165: one native encoder call per aspect ratio (chunked by ``max_bs``); returns per-view
    lists ``(x, pos)`` in input order."""
    if verbose:
        print('running encoder')
    n = true_shape.shape[0]
    device = device or true_shape.device
    out_device = "cpu" if preserve_gpu_mem else device
    shape_stacks, index_stacks, img_stacks = stack_views(true_shape, [imgs], max_bs=max_bs)
    x, pos = [None] * n, [None] * n
    for img_stack, shape_stack, idx in zip(img_stacks, shape_stacks, index_stacks):
        xs, ps = encoder(img_stack.to(device), shape_stack)
        xs, ps = xs.to(out_device), ps.to(out_device)
        for j, i in enumerate(idx):
            x[i], pos[i] = xs[j], ps[j]
    return x, pos


@torch.no_grad()
def inference_multi_ar_batch(encoder, decoder, imgs, true_shape, mem=None, verbose=False,
                             encoder_precomputed_features=None, preserve_gpu_mem=False,
                             post_process_function=_default_post_process, device=None, render=False, viser_server=None):
    """engine/inference.py:168-202: ONE decoder call (= one native ``must3r_hip_decode``) over already stacked groups.
    ``imgs`` / ``true_shape``: lists with one stacked tensor per aspect ratio.  Returns ``(mem, [per-group result])``."""
    device = device or true_shape[0].device
    out_device = "cpu" if preserve_gpu_mem else device
    if encoder_precomputed_features is None:
        x, pos = [], []
        for img_g, shape_g in zip(imgs, true_shape):
            xg, pg = encoder(img_g.to(device), shape_g)
            x.append(xg)
            pos.append(pg)
    else:
        x, pos = encoder_precomputed_features
    x = [v.unsqueeze(0).to(device) for v in x]
    pos = [v.unsqueeze(0).to(device) for v in pos]
    # the decoder reads (H, W) as host integers (head.py:33): hand the shapes over where they already are
    shapes = [v.unsqueeze(0) for v in true_shape]
    mem, pointmaps = decoder(x, pos, shapes, mem, render=render)
    out = []
    for pm in pointmaps:
        pm = pm.squeeze(0)
        if post_process_function is not None:
            pm = {k: v.to(out_device) for k, v in post_process_function(pm).items()}
        else:
            pm = pm.to(out_device)
        out.append(pm)
    return mem, out


def _update_step(encoder, decoder, x, pos, imgs, true_shape, lo, hi, mem, max_bs, device, verbose, preserve_gpu_mem,
                 post_process_function, viser_server):
    """Views lo..hi-1 update the memory: encode what is missing (engine/inference.py:266-273 / :401-408), group, decode
    (:275-284 / :410-423), hand the results back in input order (:286 / :447).  Returns (new_mem, per-view results, first
    label of the batch)."""
    sl = slice(lo, hi)
    if any(_is_missing(v) for v in x[sl]) or any(_is_missing(v) for v in pos[sl]):
        x[sl], pos[sl] = encoder_multi_ar(encoder, imgs[sl], true_shape[sl], verbose=False, max_bs=max_bs, device=device)
    shape_stacks, index_stacks, x_stacks, pos_stacks, img_stacks = stack_views(true_shape[sl], [x[sl], pos[sl], imgs[sl]],
                                                                               max_bs=max_bs)
    first_label = 0 if mem is None else int(mem[2])     # decoder.py:241-249: labels = arange(n) + mem_nimgs, group by group
    new_mem, results = inference_multi_ar_batch(encoder, decoder, img_stacks, shape_stacks, mem, verbose=verbose,
                                                encoder_precomputed_features=(x_stacks, pos_stacks),
                                                preserve_gpu_mem=preserve_gpu_mem, post_process_function=post_process_function,
                                                device=device, viser_server=viser_server)
    return new_mem, unstack_pointmaps(index_stacks, results), first_label


def _reserve(decoder, encoder, true_shape, bounds, scratch_batch):
    """Tell the native decoder how many tokens the memory will hold (all views of ``bounds``, plus the largest batch when
    refinement passes append scratch copies), so that its K|V buffers are allocated once instead of doubling."""
    if not hasattr(decoder, "reserve_memory_tokens"):
        return
    p = int(getattr(encoder, "patch_size", 16))
    tokens = [(int(h) // p) * (int(w) // p) for h, w in true_shape[:bounds[-1]].tolist()]
    per_batch = [sum(tokens[lo:hi]) for lo, hi in zip(bounds[:-1], bounds[1:])]
    decoder.reserve_memory_tokens = sum(tokens) + (max(per_batch, default=0) if scratch_batch else 0)


def _cumulative(mem_batches):
    bounds = [0]
    for nb in mem_batches:
        bounds.append(bounds[-1] + int(nb))
    return bounds


def _empty_cache_if(preserve_gpu_mem):
    if preserve_gpu_mem and torch.cuda.is_available():
        torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------------------------
# offline reconstruction: memory from the keyframes, then render everything
# ------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def inference_multi_ar(encoder, decoder, imgs, img_ids, true_shape, mem_batches, verbose=False, max_bs=None, to_render=None,
                       encoder_precomputed_features=None, precomputed_mem=None, preserve_gpu_mem=False,
                       post_process_function=_default_post_process, device=None, return_mem=False, viser_server=None,
                       num_refinements_iterations=0):
    """engine/inference.py:369-527.  ``imgs``: list of [3,H,W]; ``img_ids``: list of 0-dim tensors; ``true_shape``: list of
    [2]; the first ``sum(mem_batches)`` views build the memory batch by batch, every further pass
    (``num_refinements_iterations``) re-decodes them against the full memory and overwrites their tokens (not those of the
    reference image); then all views (or ``to_render``) are rendered against the memory, one decoder call per aspect
    ratio (chunked by ``max_bs``).  Returns ``([mem,] pointmaps_0, pointmaps)`` as lists of per-view results."""
    true_shape = torch.stack(true_shape, dim=0)
    n_views = true_shape.shape[0]
    device = device or true_shape.device
    true_shape = _host(true_shape)                       # one copy per scene; all grouping below is host arithmetic
    x, pos = ([None] * n_views, [None] * n_views) if encoder_precomputed_features is None else encoder_precomputed_features

    if precomputed_mem is None:
        if verbose:
            print('updating memory')
        mem = None
        bounds = _cumulative(mem_batches)
        pointmaps_0 = [None] * bounds[-1]
        label_of = {}                                     # image id -> label of its tokens in the memory
        _reserve(decoder, encoder, true_shape, bounds, num_refinements_iterations > 0)
        for _ in range(num_refinements_iterations + 1):
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                ids = [int(v) for v in img_ids[lo:hi]]
                refresh = all(i in label_of for i in ids)   # :413-416 (an empty batch counts as a refresh, like all([]))
                new_mem, results, first_label = _update_step(encoder, decoder, x, pos, imgs, true_shape, lo, hi, mem, max_bs,
                                                             device, verbose, preserve_gpu_mem, post_process_function,
                                                             viser_server)
                if refresh:
                    assert mem is not None
                    for j, i in enumerate(ids):
                        if label_of[i] == 0:
                            continue                       # :432-433 the reference image keeps its first tokens
                        update_in_mem(mem[0], new_mem[0], mem[1], new_mem[1], label_of[i], first_label + j)
                    del new_mem
                    rewind_mem(mem[0])                     # the appended rows were a scratch copy: the next update appends there again
                else:
                    mem = new_mem
                    for j, i in enumerate(ids):
                        label_of[i] = first_label + j      # :443-444
                pointmaps_0[lo:hi] = results
                if viser_server is not None:
                    viser_server.set_views(img_ids[lo:hi], imgs[lo:hi], results, [True] * (hi - lo))
                _empty_cache_if(preserve_gpu_mem)
    else:
        pointmaps_0 = None
        mem = precomputed_mem

    if to_render is not None:                              # :465-472
        x = [x[v] for v in to_render]
        pos = [pos[v] for v in to_render]
        true_shape = true_shape[to_render].contiguous()
        imgs = [imgs[v] for v in to_render]
        img_ids = [img_ids[v] for v in to_render]
        n_views = len(x)

    assert mem is not None
    if verbose:
        print(f"Nmem={get_Nmem(mem)}")
    if n_views == 0:
        return (mem, pointmaps_0, []) if return_mem else (pointmaps_0, [])
    if verbose:
        print(f'rendering {n_views} extra images')
    shape_stacks, index_stacks, x_stacks, pos_stacks, img_stacks, id_stacks = stack_views(true_shape, [x, pos, imgs, img_ids],
                                                                                          max_bs=max_bs)
    rendered = []
    for x_g, pos_g, shape_g, img_g, id_g in zip(x_stacks, pos_stacks, shape_stacks, img_stacks, id_stacks):
        feats = None if (x_g is None or pos_g is None) else ([x_g], [pos_g])
        _, res = inference_multi_ar_batch(encoder, decoder, [img_g], [shape_g], mem, verbose=verbose,
                                          encoder_precomputed_features=feats, preserve_gpu_mem=preserve_gpu_mem,
                                          post_process_function=post_process_function, device=device, render=True,
                                          viser_server=viser_server)
        rendered.append(res[0])
        if viser_server is not None:
            per_view = unstack_pointmaps([list(range(id_g.shape[0]))], res)
            for j in range(id_g.shape[0]):
                viser_server.set_views([id_g[j]], [img_g[j]], [per_view[j]])
    pointmaps = unstack_pointmaps(index_stacks, rendered)
    return (mem, pointmaps_0, pointmaps) if return_mem else (pointmaps_0, pointmaps)


# ------------------------------------------------------------------------------------------------------------------
# online: every frame updates the memory, keyframes stay, the rest lives for local_context_size frames
# ------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def inference_video_multi_ar(encoder, decoder, imgs, true_shape, mem_batches, verbose=False, max_bs=None,
                             encoder_precomputed_features=None, preserve_gpu_mem=False,
                             post_process_function=_default_post_process, device=None, return_mem=False, viser_server=None,
                             num_refinements_iterations=0, local_context_size=25,
                             is_keyframe_function=lambda id, res, scene_state: (id % 3 == 0), scene_state=None,
                             scene_state_update_function=lambda res, scene_state: scene_state):
    """engine/inference.py:231-366.  Frame ids are the positions in ``imgs``.  All frames of the first batch are keyframes
    (:294-301); a later frame is one if ``is_keyframe_function(id, result, scene_state)`` says so (:312).  On a further pass
    a keyframe's new tokens overwrite its old ones and the appended copy is dropped (:315-320), a non-keyframe's appended
    tokens take its old label back (:321-324).  Non-keyframes leave the memory ``local_context_size`` frames later
    (:336-339) and at the end of every pass (:354-359).  Returns ``pointmaps_0`` (``(mem, pointmaps_0)`` with ``return_mem``;
    ``mem`` is a list like the reference's)."""
    true_shape = torch.stack(true_shape, dim=0)
    n_views = true_shape.shape[0]
    device = device or true_shape.device
    true_shape = _host(true_shape)                       # one copy per scene; all grouping below is host arithmetic
    x, pos = ([None] * n_views, [None] * n_views) if encoder_precomputed_features is None else encoder_precomputed_features
    if verbose:
        print('updating memory')
    mem = None
    bounds = _cumulative(mem_batches)
    pointmaps_0 = [None] * bounds[-1]
    label_of, keyframes = {}, set()
    img_ids = [torch.tensor(v) for v in range(n_views)]

    def evict(frame_id):
        if frame_id not in keyframes:
            mem[0], mem[1] = remove_from_mem(mem[0], mem[1], label_of[frame_id])

    for _ in range(num_refinements_iterations + 1):
        window = deque()
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            ids = list(range(lo, hi))
            new_mem, results, first_label = _update_step(encoder, decoder, x, pos, imgs, true_shape, lo, hi, mem, max_bs, device,
                                                         verbose, preserve_gpu_mem, post_process_function, viser_server)
            pointmaps_0[lo:hi] = results
            mem = list(new_mem)
            flags = []
            if not label_of:                                  # initialisation: everything is a keyframe
                for j, i in enumerate(ids):
                    label_of[i] = first_label + j
                    window.append(i)
                    keyframes.add(i)
                    flags.append(True)
                    scene_state = scene_state_update_function(results[j], scene_state)
            else:
                for j, i in enumerate(ids):
                    seen = i in label_of
                    is_key = (i in keyframes) if seen else is_keyframe_function(i, results[j], scene_state)
                    window.append(i)
                    flags.append(is_key)
                    new_label = first_label + j
                    if seen and is_key:
                        if label_of[i] != 0:                  # :318 the reference image is not refreshed
                            mem[0] = update_in_mem(mem[0], mem[0], mem[1], mem[1], label_of[i], new_label)
                        mem[0], mem[1] = remove_from_mem(mem[0], mem[1], new_label)
                    elif seen:
                        mem[1] = restore_label_in_mem(mem[1], label_of[i], new_label)
                    else:
                        label_of[i] = new_label
                        if is_key:
                            keyframes.add(i)
                            scene_state = scene_state_update_function(results[j], scene_state)
            if viser_server is not None:
                viser_server.set_views(img_ids[lo:hi], imgs[lo:hi], results, flags)
            while len(window) > local_context_size:
                evict(window.popleft())
            mem[2] = len(label_of)                            # :342 labels keep counting from the number of distinct frames
            _empty_cache_if(preserve_gpu_mem)
        assert mem is not None
        while window:                                         # :354-359 leave only the keyframes for the next pass
            evict(window.popleft())
    return (mem, pointmaps_0) if return_mem else pointmaps_0


# ------------------------------------------------------------------------------------------------------------------
# single aspect ratio, batched tensors (the reference's evaluation / training-time driver, no-grad use)
# ------------------------------------------------------------------------------------------------------------------
def inference_encoder(encoder, imgs, true_shape_view, max_bs=None, requires_grad=False):
    """engine/inference.py:570-592.  ``imgs`` [B, n, 3, H, W], ``true_shape_view`` [B*n, 2] -> ``x`` [B, n, N, C],
    ``pos`` [B, n, N, 2]; in slices of ``max_bs`` images when asked."""
    if requires_grad:
        raise NotImplementedError("the native encoder is inference-only (engine/train.py is out of scope)")
    with torch.no_grad():
        B, n = imgs.shape[:2]
        flat = imgs.reshape(B * n, *imgs.shape[2:])
        if max_bs is None or B * n <= max_bs:
            x, pos = encoder(flat, true_shape_view)
        else:
            parts = [encoder(a, b) for a, b in zip(torch.split(flat, max_bs), torch.split(true_shape_view, max_bs))]
            x = torch.cat([p[0] for p in parts])
            pos = torch.cat([p[1] for p in parts])
        return x.view(B, n, *x.shape[1:]), pos.view(B, n, *pos.shape[1:])


@torch.no_grad()
def inference(encoder, decoder, imgs, true_shape, mem_batches, verbose=False, max_bs=None, train_decoder_skip=0,
              to_render=None, encoder_requires_grad=False):
    """engine/inference.py:595-688.  ``imgs`` [B, n, 3, H, W], ``true_shape`` [B, n, 2] (one aspect ratio).  Returns
    ``(pointmaps_0 [B, sum(mem_batches) - skipped, H, W, 7], pointmaps [B, n_rendered, H, W, 7])``.  The first
    ``train_decoder_skip`` batches update the memory without contributing to ``pointmaps_0`` (:610-617)."""
    B, n = imgs.shape[:2]
    x, pos = inference_encoder(encoder, imgs, true_shape.view(B * n, 2), max_bs, encoder_requires_grad)
    N, C = x.shape[2:]
    bounds = _cumulative(mem_batches)
    mem, first_pass, out_shape = None, [], None
    for b, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        mem, pm = decoder(x[:, lo:hi].contiguous(), pos[:, lo:hi].contiguous(), true_shape[:, lo:hi].contiguous(), mem,
                          render=False)
        out_shape = out_shape or pm.shape
        if b >= train_decoder_skip:
            first_pass.append(pm)
    if first_pass:
        pointmaps_0 = torch.cat(first_pass, dim=1)
    else:
        pointmaps_0 = torch.empty((B, 0, *out_shape[2:]), dtype=x.dtype, device=x.device)
    if to_render is not None:
        x, pos, true_shape = x[:, to_render].contiguous(), pos[:, to_render].contiguous(), true_shape[:, to_render].contiguous()
        n = x.shape[1]
    assert mem is not None
    mem_vals, mem_labels, mem_nimgs, mem_prot_imgs, mem_prot_tok = mem
    if verbose:
        print(f"Nmem={mem_vals[-1].shape[1]}")
    if n == 0:
        return pointmaps_0, torch.empty((B, 0, *pointmaps_0.shape[2:]), dtype=x.dtype, device=x.device)
    if max_bs is None or B * n <= max_bs:
        _, pointmaps = decoder(x, pos, true_shape, mem, render=True)
        return pointmaps_0, pointmaps
    # slice by slice (:661-686): every image becomes a batch element of its own with its scene's memory
    owner_scene = torch.arange(B, device=x.device).repeat_interleave(n)
    parts = []
    for idx in torch.split(torch.arange(B * n, device=x.device), max_bs):
        scene = owner_scene[idx]
        mem_slice = ([v[scene] for v in mem_vals], mem_labels[scene], mem_nimgs, mem_prot_imgs, mem_prot_tok)
        _, pm = decoder(x.view(B * n, 1, N, C)[idx], pos.view(B * n, 1, N, 2)[idx], true_shape.view(B * n, 1, 2)[idx], mem_slice,
                        render=True)
        parts.append(pm.squeeze(1))
    pointmaps = torch.cat(parts)
    return pointmaps_0, pointmaps.view(B, n, *pointmaps.shape[1:])
