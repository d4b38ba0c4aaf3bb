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
    if world == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no CUDA all_gather: stage through the host (only used when ranks share a GPU in tests)
        return all_gather_varlen(t.cpu(), group, counts).to(t.device)
    if counts is None:
        counts = gather_counts(t.shape[0], t.device, group)
    assert counts[rank] == t.shape[0], (counts, rank, t.shape)
    kmax = max(counts)
    if kmax == 0:
        return t
    if min(counts) == kmax:
        pad = t.contiguous()                      # equal shards: no staging copy
    else:
        pad = torch.zeros((kmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


@torch.no_grad()
def run_scene_sharded(encoder, decoder, imgs_local, true_shape_local, keyframe_local, group=None, comm_dtype=None,
                      mem_batches=None, gather_outputs=False):
    """One scene whose views are sharded over the ranks of ``group``.

    imgs_local [v,3,H,W], true_shape_local [v,2]: this rank's views (global order = rank order, then local order).
    keyframe_local bool[v]: which local views enter the memory.  Returns dict(render=[v,H,W,7] local pointmaps,
    mem=memory tuple (identical on all ranks), n_keyframes) (+ render_all when gather_outputs).
    """
    x, pos = encoder(imgs_local, true_shape_local)
    kf = keyframe_local.to(x.device)
    kx = x[kf]
    if comm_dtype is not None:
        kx = kx.to(comm_dtype)  # 16-bit on the wire; the decoder rounds its operands to this type anyway
    counts = gather_counts(int(keyframe_local.sum()), x.device, group)       # host-known locally: one exchange for all three
    kx = all_gather_varlen(kx, group, counts).float()
    kpos = all_gather_varlen(pos[kf], group, counts)
    kts = all_gather_varlen(true_shape_local.to(x.device)[kf], group, counts).cpu()   # host copy: no per-call sync in the decoder
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
    if x.shape[0] > 0:
        _, pm = decoder(x.unsqueeze(0), pos.unsqueeze(0), true_shape_local.cpu().unsqueeze(0), mem, render=True)
        out["render"] = pm[0]
    else:
        out["render"] = x.new_zeros((0,))
    if gather_outputs:
        out["render_all"] = all_gather_varlen(out["render"], group)
    return out
