"""TEST INFRASTRUCTURE ONLY.  CPU stand-in for the context-parallel decoder call (``MUSt3R.forward(.., cp=ContextParallel)``; include/must3r_hip.h ``must3r_hip_cp``).

The same protocol as the native path, restated on CPU tensors with ``torch.distributed`` (gloo in tests/test_parallel_gloo.py): a one-view memory update against a
memory that is SHARDED over the ranks -- every rank holds some rows of every layer's K|V memory -- computes, per decoder block, the flash-attention PARTIAL of its own
rows (m = row max of the scaled scores in the log2 domain, l = sum 2^(s - m), O = sum 2^(s - m) v), all-gathers the partials and merges them
(O = sum_r 2^(m_r - m*) O_r / sum_r 2^(m_r - m*) l_r); everything else of the block is replicated.  Semantics of the block: ``oracle/must3r_ref.py::decoder_block``
(CachedDecoderBlock.forward, must3r/model/blocks/layers.py:90-99; CachedCrossAttention.forward, must3r/model/blocks/attention.py:139-149), memory algebra:
``decoder_forward`` (must3r/model/decoder.py:267-350) for the lone-view update case (the view's own new tokens are excluded by make_mem_mask, decoder.py:119-139, so
the keys are exactly the old memory).
"""
import math

import torch
import torch.distributed as dist

from . import must3r_ref as R

LOG2E = 1.4426950408889634


def _partial(q, k, v):
    """q [1,H,N,64]; k, v [1,H,M,64] (M may be 0) -> (m [H,N], l [H,N], O [H,N,64]) in the log2 domain with scale 1/sqrt(64)."""
    H, N = q.shape[1], q.shape[2]
    if k.shape[2] == 0:
        return torch.full((H, N), -math.inf), torch.zeros((H, N)), torch.zeros((H, N, 64))
    s = (q[0] @ k[0].transpose(-2, -1)) * (q.shape[-1] ** -0.5) * LOG2E
    m = s.amax(dim=-1)
    p = torch.exp2(s - m[..., None])
    return m, p.sum(-1), p @ v[0]


def _merge(parts):
    m = torch.stack([p[0] for p in parts]).amax(dim=0)
    w = [torch.exp2(p[0] - m) for p in parts]
    L = sum(wi * p[1] for wi, p in zip(w, parts))
    O = sum(wi[..., None] * p[2] for wi, p in zip(w, parts))
    return O / L[..., None]


def _all_gather_partial(part, group):
    if not (dist.is_available() and dist.is_initialized()):
        return [part]
    world = dist.get_world_size(group)
    flat = torch.cat([part[0].reshape(-1), part[1].reshape(-1), part[2].reshape(-1)])
    out = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(out, flat, group=group)
    H, N = part[0].shape
    return [(o[:H * N].view(H, N), o[H * N:2 * H * N].view(H, N), o[2 * H * N:].view(H, N, 64)) for o in out]


def decoder_forward_cp(sd, cfg, x, pos, true_shape, mem_local, cp, memory_mode="kv"):
    """One-view memory update with the memory sharded over ``cp.group``: ``mem_local`` = this rank's memory tuple (its rows, possibly none).  Returns
    ``(mem_tuple with the new view's rows appended to the LOCAL rows, pointmaps [1,1,H,W,7])`` -- the caller decides which rank keeps the rows."""
    assert memory_mode == "kv" and x.shape[0] == 1 and x.shape[1] == 1 and cp.n_mem_total > 0
    D, H = cfg.dec_dim, cfg.dec_heads
    mem_vals, mem_labels, mem_nimgs, _, _ = mem_local
    mem_vals = [m.float() for m in mem_vals]
    n, N = 1, x.shape[2]
    cur = R.linear(x.reshape(n, N, -1).float(), sd["feat_embed_enc_to_dec.weight"], sd["feat_embed_enc_to_dec.bias"]) + sd["image2_embed"].view(1, 1, D)
    p = pos.reshape(n, N, 2)
    new_mem = []
    for l in range(cfg.dec_depth):
        b = f"blocks_dec.{l}"
        new_mem.append(cur.reshape(1, -1, D))
        h = R.layer_norm(cur, sd[b + ".norm1.weight"], sd[b + ".norm1.bias"], 1e-6)
        xx = cur + R.self_attention(sd, b + ".attn", h, p, H, cfg)
        y = mem_vals[l]
        key, value = y[..., :D], y[..., D:]
        h = R.layer_norm(xx, sd[b + ".norm2.weight"], sd[b + ".norm2.bias"], 1e-6)
        q = R.split_heads(R.linear(h, sd[b + ".cross_attn.projq.weight"], sd[b + ".cross_attn.projq.bias"]), H)
        part = _partial(q, R.split_heads(key, H), R.split_heads(value, H))
        o = R.merge_heads(_merge(_all_gather_partial(part, cp.group)).unsqueeze(0))
        xx = xx + R.linear(o, sd[b + ".cross_attn.proj.weight"], sd[b + ".cross_attn.proj.bias"])
        h = R.layer_norm(xx, sd[b + ".norm3.weight"], sd[b + ".norm3.bias"], 1e-6)
        cur = xx + R.mlp(sd, b + ".mlp", h)
    if "feedback_layer.fc1.weight" in sd:
        fb = R.layer_norm(new_mem[-1], sd["feedback_norm.weight"], sd["feedback_norm.bias"], 1e-5)
        off = R.mlp(sd, "feedback_layer", fb)
        new_mem = [m + off for m in new_mem[:-1]] + [new_mem[-1]]
    elif "feedback_layer.weight" in sd:
        fb = R.layer_norm(new_mem[-1], sd["feedback_norm.weight"], sd["feedback_norm.bias"], 1e-5)
        off = R.linear(fb, sd["feedback_layer.weight"], sd["feedback_layer.bias"])
        new_mem = [m + off for m in new_mem[:-1]] + [new_mem[-1]]
    mem_out = [torch.cat((mem_vals[l], R.prepare_y(sd, f"blocks_dec.{l}", new_mem[l], "kv")), dim=1) for l in range(cfg.dec_depth)]
    labels = torch.cat([mem_labels, torch.full((1, N), mem_nimgs, dtype=torch.int64)], dim=1)
    Hh, Ww = (int(v) for v in true_shape.reshape(-1, 2)[0])
    pm = R.unpatchify_head(sd, cfg, cur, Hh, Ww).unsqueeze(0)
    tot = mem_nimgs + 1
    return (mem_out, labels, tot, tot, labels.shape[1]), pm
