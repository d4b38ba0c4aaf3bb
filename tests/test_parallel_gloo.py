"""CPU, world_size 2, gloo: the multi-GPU scene driver (must3r_amd/parallel.py) with the oracle standing in for the
two forwards -- the sharded result must equal the single-process result bit for bit (the replicated memory update
is deterministic), and the variable-length all-gather must keep rank order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from must3r_amd.config import TINY
from must3r_amd import synthetic as S
from must3r_amd.parallel import all_gather_varlen, run_scene_sharded, run_video_sharded, shard_range

V, H, W = 6, 48, 64


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_pair():
    from oracle import must3r_ref as R
    cfg = TINY
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    enc = lambda img, ts: R.encoder_forward(sde, cfg, img, ts)  # noqa: E731
    dec = lambda x, pos, ts, mem=None, render=False: R.decoder_forward(sdd, cfg, x, pos, ts, mem, render, "kv")  # noqa: E731
    return enc, dec


def _oracle_pair_cp():
    """the oracle pair whose decoder also takes ``cp=`` (oracle/cp_ref.py: the context-parallel protocol on CPU tensors)"""
    from oracle import must3r_ref as R, cp_ref as CP
    cfg = TINY
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    enc = lambda img, ts: R.encoder_forward(sde, cfg, img, ts)  # noqa: E731

    def dec(x, pos, ts, mem=None, render=False, cp=None):
        if cp is not None:
            return CP.decoder_forward_cp(sdd, cfg, x, pos, ts, mem, cp)
        return R.decoder_forward(sdd, cfg, x, pos, ts, mem, render, "kv")
    return enc, dec


def _keyframes():
    return torch.tensor([True, False, True, True, False, True])  # 4 keyframes, unevenly spread over the 2 shards


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        # varlen gather keeps rank order, handles empty shards
        t = torch.arange(rank * 10, rank * 10 + (3 if rank == 0 else 1)).float().view(-1, 1)
        g = all_gather_varlen(t)
        assert g.flatten().tolist() == [0.0, 1.0, 2.0, 10.0]
        e = all_gather_varlen(torch.zeros((0, 2)) if rank == 1 else torch.ones((2, 2)))
        assert e.shape == (2, 2)
        # r05: ranks holding DIFFERENT stored shapes with the same token count -- every gathered keyframe's positions must be those of
        # ITS OWN shape (rebuilt from the true shape that travels with the tokens), i.e. exactly what gathering the positions themselves gives
        from must3r_amd.parallel import _gather_keyframes, grid_positions
        hw, nloc = ((48, 64), 2) if rank == 0 else ((64, 48), 1)
        xk = torch.randn(nloc, 12, 8)
        posk = grid_positions(hw[0] // 16, hw[1] // 16, torch.device("cpu")).unsqueeze(0).expand(nloc, -1, -1).contiguous()
        tsk = torch.tensor([hw] * nloc)
        allk = torch.ones(nloc, dtype=torch.bool)
        _, kpos_rebuilt, kts_a = _gather_keyframes(xk, posk, tsk, allk, None, None, None, 16, False)
        _, kpos_gathered, kts_b = _gather_keyframes(xk, posk, tsk, allk, None, None, None, None, False)
        assert torch.equal(kpos_rebuilt, kpos_gathered) and torch.equal(kts_a, kts_b) and kts_a.tolist() == [[48, 64], [48, 64], [64, 48]]
        assert not torch.equal(kpos_rebuilt[0], kpos_rebuilt[2])
        with pytest.raises(ValueError, match="does not fit the keyframe payload"):   # the payload row holds two base-256 digits per extent
            _gather_keyframes(xk, posk, torch.tensor([[65536, 16]] * nloc), allk, None, None, None, 16, False)
        with pytest.raises(ValueError, match="does not have the 12 tokens"):        # a shape that is not this stack's token count
            _gather_keyframes(xk, posk, torch.tensor([[64, 64]] * nloc), allk, None, None, None, 16, False)
        imgs, ts = S.make_images(V, H, W, 0)
        lo, hi = shard_range(V, rank, world)
        enc, dec = _oracle_pair()
        with torch.no_grad():
            out = run_scene_sharded(enc, dec, imgs[lo:hi], ts[lo:hi], _keyframes()[lo:hi], gather_outputs=True)
            # the same with the keyframe counts of both ranks known on the host (no count exchange): identical result
            kc = [int(_keyframes()[a:b].sum()) for a, b in (shard_range(V, r, world) for r in range(world))]
            out_s = run_scene_sharded(enc, dec, imgs[lo:hi], ts[lo:hi], _keyframes()[lo:hi], gather_outputs=True, keyframe_counts=kc)
            assert torch.equal(out_s["render_all"], out["render_all"]) and torch.equal(out_s["mem"][0][-1], out["mem"][0][-1])
        torch.save({"render_all": out["render_all"], "mem_last": out["mem"][0][-1], "labels": out["mem"][1], "K": out["n_keyframes"]},
                   os.path.join(out_dir, f"r{rank}.pt"))
        # a rank WITHOUT views (fewer views than ranks): empty encoder batch skipped, [0,H,W,7] render, gather still works
        lo1, hi1 = shard_range(1, rank, world)
        with torch.no_grad():
            o1 = run_scene_sharded(enc, dec, imgs[lo1:hi1], ts[lo1:hi1], torch.ones(hi1 - lo1, dtype=torch.bool), gather_outputs=True)
        assert o1["render"].shape == (hi1 - lo1, H, W, 7) and o1["render_all"].shape == (1, H, W, 7) and o1["n_keyframes"] == 1
        # streaming schedule over the sharded sequence (window 3, every 2nd frame a keyframe)
        with torch.no_grad():
            ov = run_video_sharded(enc, dec, imgs[lo:hi], ts[lo:hi], local_context_size=3, is_keyframe=lambda i: i % 2 == 0,
                                   gather_outputs=True)
        torch.save({"render_all": ov["render_all"], "pm0": ov["pointmaps_0"], "labels": ov["mem"][1], "kf": ov["keyframes"],
                    "mem_last": ov["mem"][0][-1]}, os.path.join(out_dir, f"v{rank}.pt"))
        # r06 (SURVEY.md section 8f "later"): the same stream with the MEMORY sharded over the ranks and the per-frame cross attention context-parallel
        enc_c, dec_c = _oracle_pair_cp()
        with torch.no_grad():
            oc = run_video_sharded(enc_c, dec_c, imgs[lo:hi], ts[lo:hi], local_context_size=3, is_keyframe=lambda i: i % 2 == 0,
                                   gather_outputs=True, context_parallel=True)
        assert oc["rows_per_rank"][rank] == int(oc["mem_local"][1].shape[1]) and sum(oc["rows_per_rank"]) == int(oc["mem"][1].shape[1])
        assert set(oc["mem_local"][1].flatten().tolist()) <= {lab for lab in range(V) if lab % world == rank}, "a rank holds only its own labels"
        torch.save({"render_all": oc["render_all"], "pm0": oc["pointmaps_0"], "labels": oc["mem"][1], "kf": oc["keyframes"], "mem_last": oc["mem"][0][-1],
                    "rows": oc["rows_per_rank"], "local_labels": oc["mem_local"][1]}, os.path.join(out_dir, f"c{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n in (0, 1, 5, 20, 21):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_sharded_scene_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["K"] == 4 and torch.equal(r0["render_all"], r1["render_all"]) and torch.equal(r0["mem_last"], r1["mem_last"])
    # single process reference: same keyframes in global order, then render everything
    imgs, ts = S.make_images(V, H, W, 0)
    enc, dec = _oracle_pair()
    torch.set_num_threads(2)
    with torch.no_grad():
        one = run_scene_sharded(enc, dec, imgs, ts, _keyframes())
    assert torch.allclose(one["render"], r0["render_all"], atol=1e-6)
    assert torch.allclose(one["mem"][0][-1], r0["mem_last"], atol=1e-6) and torch.equal(one["mem"][1], r0["labels"])
    # streaming: sharded == single process (engine.run_video + render of every frame against the final memory)
    from must3r_amd.engine import run_video
    v0, v1 = torch.load(tmp_path / "v0.pt"), torch.load(tmp_path / "v1.pt")
    assert torch.equal(v0["render_all"], v1["render_all"]) and torch.equal(v0["pm0"], v1["pm0"]) and v0["kf"] == v1["kf"]
    with torch.no_grad():
        memv, pm0, kf = run_video(enc, dec, imgs, ts, local_context_size=3, is_keyframe=lambda i: i % 2 == 0)
        _, renv = dec(*[t.unsqueeze(0) for t in enc(imgs, ts)], ts.unsqueeze(0), memv, render=True)
    assert kf == v0["kf"] and torch.equal(memv[1], v0["labels"])
    assert torch.allclose(pm0, v0["pm0"], atol=1e-6) and torch.allclose(renv[0], v0["render_all"], atol=1e-6)
    assert torch.allclose(memv[0][-1], v0["mem_last"], atol=1e-6)
    # context-parallel stream (memory sharded, label L on rank L % 2): identical on both ranks, equal to the single-process stream up to the order the
    # partial sums are merged in -- the gathered memory holds the same rows per label, in rank order
    c0, c1 = torch.load(tmp_path / "c0.pt"), torch.load(tmp_path / "c1.pt")
    assert torch.equal(c0["render_all"], c1["render_all"]) and torch.equal(c0["pm0"], c1["pm0"]) and c0["kf"] == c1["kf"] == kf and c0["rows"] == c1["rows"]
    assert torch.allclose(pm0, c0["pm0"], atol=2e-5) and torch.allclose(renv[0], c0["render_all"], atol=2e-5)
    assert sorted(c0["labels"].flatten().tolist()) == sorted(memv[1].flatten().tolist())
    assert sum(c0["rows"]) == memv[1].shape[1] and min(c0["rows"]) > 0, c0["rows"]          # both ranks really hold part of the memory
    for lab in set(memv[1].flatten().tolist()):   # the rows of every surviving label are those of the single-process memory
        a = memv[0][-1][0][memv[1][0] == lab]
        b = c0["mem_last"][0][c0["labels"][0] == lab]
        assert torch.allclose(a, b, atol=2e-5), lab


def test_positions_rebuilt_from_the_grid_equal_the_encoders():
    """r04: positions never travel between ranks -- `grid_positions` must be exactly what the encoder returns for a view (croco
    PositionGetter, SURVEY.md Appendix A: row-major (y, x))."""
    from oracle import must3r_ref as R
    from must3r_amd.parallel import grid_positions
    sde = S.make_encoder_state_dict(TINY, 0)
    for h, w in ((48, 64), (64, 64), (32, 96)):
        imgs, ts = S.make_images(1, h, w, 0)
        _, pos = R.encoder_forward(sde, TINY, imgs, ts)
        assert torch.equal(grid_positions(h // 16, w // 16, torch.device("cpu")), pos[0])
