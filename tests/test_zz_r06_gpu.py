"""GPU tests added in round 6: the context-parallel cross attention over a memory sharded across ranks (SURVEY.md section 8f "later";
include/must3r_hip.h ``must3r_hip_cp``; must3r_amd/parallel.py ``run_video_sharded(context_parallel=True)``) -- through RCCL with a process group
of one rank, and with TWO processes that share the box's GPU (gloo, partials staged through the host): both must reproduce the single-process
stream within the precision mode's tolerance and the oracle's."""
import os
import socket

import pytest
import torch

from must3r_amd import synthetic as S
from must3r_amd.config import SMALL, TINY
from util import TOL, rel_inf
from test_model_gpu import build
from test_ops_gpu import record

pytestmark = pytest.mark.gpu

V, H, W = 7, 224, 224
WINDOW, KF = 3, (lambda i: i % 2 == 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_stream(cfg, imgs, ts):
    """the fp32 CPU oracle of the same stream: engine.run_video with the oracle standing in for the two forwards + the render of every frame"""
    from oracle import must3r_ref as R
    from must3r_amd.engine import run_video
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    enc = lambda img, t: R.encoder_forward(sde, cfg, img, t)  # noqa: E731
    dec = lambda x, pos, t, mem=None, render=False: R.decoder_forward(sdd, cfg, x, pos, t, mem, render, "kv")  # noqa: E731
    with torch.no_grad():
        mem, pm0, kf = run_video(enc, dec, imgs, ts, local_context_size=WINDOW, is_keyframe=KF)
        x, pos = enc(imgs, ts)
        _, ren = dec(x.unsqueeze(0), pos.unsqueeze(0), ts.unsqueeze(0), mem, render=True)
    return pm0, ren[0], kf


@pytest.mark.parametrize("partials", [True, "fp32"])
def test_context_parallel_stream_through_rccl_group_of_one_rank(partials):
    """One rank, RCCL: every layer's partial really goes through all_gather_into_tensor (12 exchanges per one-view call), the merge of ONE slot reproduces the
    single-process stream (same sums; the local attention is always split in two or more, so the first frames -- whose memory the plain path attends unsplit --
    differ in the last bits), and both sit inside the mode's tolerance of the oracle."""
    import torch.distributed as dist
    from must3r_amd.engine import run_video
    from must3r_amd.parallel import run_video_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg, prec = SMALL, "fp16w2"
        enc, dec = build(cfg, prec)
        imgs, ts = S.make_images(V, H, W, 11)
        ov = run_video_sharded(enc, dec, imgs.cuda(), ts, local_context_size=WINDOW, is_keyframe=KF, frame_counts=[V], context_parallel=partials, gather_outputs=True)
        memv, pm0, kfs = run_video(enc, dec, imgs.cuda(), ts, local_context_size=WINDOW, is_keyframe=KF)
        torch.cuda.synchronize()
        assert ov["keyframes"] == kfs and torch.equal(ov["mem"][1], memv[1]) and ov["rows_per_rank"] == [int(memv[1].shape[1])]
        assert ov["cp_exchanges"] == cfg.dec_depth * (V - 2), ov["cp_exchanges"]
        e_cp = rel_inf(ov["pointmaps_0"].cpu(), pm0.cpu())
        pm_o, ren_o, kf_o = _oracle_stream(cfg, imgs, ts)
        e_or = max(rel_inf(ov["pointmaps_0"].cpu(), pm_o), rel_inf(ov["render_all"].cpu(), ren_o))
        record("context_parallel_world1", partials=str(partials), vs_plain_stream=e_cp, vs_oracle=e_or, exchanges=ov["cp_exchanges"], bytes_gathered=ov["cp_bytes_gathered"])
        # (measured 2.7e-4 from the plain stream: the context-parallel call always leaves 16-bit split-KV partials, the plain one-view call attends these short
        # memories unsplit -- the rounding of one more 16-bit intermediate; 4.5e-4 from the oracle)
        assert kf_o == kfs and e_cp < 0.5 * TOL[prec] and e_or < TOL[prec], (e_cp, e_or)
    finally:
        dist.destroy_process_group()


def _worker2(rank, world, port, out_dir):
    import torch.distributed as dist
    from must3r_amd.parallel import run_video_sharded, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)     # the two ranks share cuda:0: the collectives go through the host
    try:
        torch.cuda.set_device(0)
        cfg, prec = SMALL, "fp16w2"
        enc, dec = build(cfg, prec)
        imgs, ts = S.make_images(V, H, W, 11)
        lo, hi = shard_range(V, rank, world)
        ov = run_video_sharded(enc, dec, imgs[lo:hi].cuda(), ts[lo:hi], local_context_size=WINDOW, is_keyframe=KF, context_parallel=True, gather_outputs=True)
        torch.cuda.synchronize()
        assert set(ov["mem_local"][1].flatten().tolist()) <= {lab for lab in range(V) if lab % world == rank}
        torch.save({"pm0": ov["pointmaps_0"].cpu(), "render_all": ov["render_all"].cpu(), "kf": ov["keyframes"], "rows": ov["rows_per_rank"],
                    "labels": ov["mem"][1].cpu(), "exchanges": ov["cp_exchanges"]}, os.path.join(out_dir, f"cp{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_context_parallel_stream_two_ranks_share_the_gpu(tmp_path):
    """World size 2 on ONE GPU: each process holds the rows of its own labels only (label L on rank L % 2), attends them, and the partials of the two processes
    are merged on both -- identical results on both ranks, the single-process stream within the precision mode's tolerance, the oracle within TOL."""
    import torch.multiprocessing as mp
    from must3r_amd.engine import run_video
    world = 2
    mp.spawn(_worker2, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    c0, c1 = torch.load(tmp_path / "cp0.pt"), torch.load(tmp_path / "cp1.pt")
    assert torch.equal(c0["pm0"], c1["pm0"]) and torch.equal(c0["render_all"], c1["render_all"]) and c0["kf"] == c1["kf"] and c0["rows"] == c1["rows"]
    assert min(c0["rows"]) > 0 and c0["exchanges"] == SMALL.dec_depth * (V - 2)
    cfg, prec = SMALL, "fp16w2"
    enc, dec = build(cfg, prec)
    imgs, ts = S.make_images(V, H, W, 11)
    memv, pm0, kfs = run_video(enc, dec, imgs.cuda(), ts, local_context_size=WINDOW, is_keyframe=KF)
    x, pos = enc(imgs.cuda(), ts)
    _, ren = dec(x.unsqueeze(0), pos.unsqueeze(0), ts.unsqueeze(0), memv, render=True)
    torch.cuda.synchronize()
    assert kfs == c0["kf"] and sorted(c0["labels"].flatten().tolist()) == sorted(memv[1].cpu().flatten().tolist())
    e_cp = max(rel_inf(c0["pm0"], pm0.cpu()), rel_inf(c0["render_all"], ren[0].cpu()))
    pm_o, ren_o, _ = _oracle_stream(cfg, imgs, ts)
    e_or = max(rel_inf(c0["pm0"], pm_o), rel_inf(c0["render_all"], ren_o))
    record("context_parallel_world2_shared_gpu", vs_plain_stream=e_cp, vs_oracle=e_or, rows_per_rank=c0["rows"])
    assert e_cp < 0.5 * TOL[prec] and e_or < TOL[prec], (e_cp, e_or)


# ------------------------------------------------------------------------------------------------------------------------------------------------
# r06: the LN fold on the chip-filling 256 x 256 tiles (GemmArgs::fold256; include/must3r_hip.h must3r_hip_op_gemm_fold256)
# ------------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [768, 1024])
def test_gemm_fold256_producers_and_consumers(D):
    """The fold of must3r/model/blocks/layers.py:51-54 / :90-99's pre-LN residual blocks on the chip-filling kernels: a residual GEMM (split weights with the sparse low
    part = proj, plain weights = fc2) leaves x - shift in fp16 plus (sum, sum of squares) per row and 64-column wave tile; the Linear that follows (split: projq, qkv
    with RoPE; plain: fc1 with GELU) multiplies the raw rows by gamma (.) W and normalises after the product.  Checked against fp64 and against the unfolded HIP route
    (LayerNorm kernel + the same GEMM kernel family), decoder width (12 slots per row) and encoder width (16)."""
    import ctypes as C
    import math
    from must3r_amd import _lib as lib
    from oracle import must3r_ref as R
    from test_ops_gpu import P, stream, _split_w
    L = lib.load()
    M, heads = 256 * 84, D // 64
    g = torch.Generator(device="cuda").manual_seed(5 + D)
    x0 = torch.randn((M, D), device="cuda", generator=g) * (1.0 + 3.0 * torch.rand((M, 1), device="cuda", generator=g)) + 0.7 * torch.randn((M, 1), device="cuda", generator=g)
    x0[:8] += 40.0          # rows whose mean is ~20 standard deviations: what the per-row shift is for
    x0[8:16, :3] *= 300.0   # rows with a few massive channels

    def sparse(Wf):
        N, K = Wf.shape
        vals = torch.empty((K // 64, N, 32), device="cuda", dtype=torch.float16)
        idx = torch.empty((K // 64, N // 32, 64), device="cuda", dtype=torch.int32)
        lib.check(L.must3r_hip_op_sparse24_pack(P(Wf.contiguous()), N, K, P(vals), P(idx), stream()))
        return vals, idx

    shift0 = x0.mean(1).contiguous()
    errs = {}
    for prod, Kp in (("proj/split", D), ("fc2/plain", 4 * D)):
        split = prod.endswith("split")
        a = torch.randn((M, Kp), device="cuda", generator=g).half()
        Wp = torch.randn((D, Kp), device="cuda", generator=g) / math.sqrt(Kp)
        bp = torch.randn((D,), device="cuda", generator=g)
        x = x0.clone()
        x16 = torch.full((M, D), float("nan"), device="cuda", dtype=torch.float16)
        cp = torch.full((M, D), float("nan"), device="cuda")
        st = torch.full((M, D // 64, 2), float("nan"), device="cuda")
        shift = shift0.clone()
        if split:
            vals, idx = sparse(Wp)
            lib.check(L.must3r_hip_op_gemm_fold256(lib.EPI_RESID_F32, 2, P(a), P(_split_w(Wp)), P(vals), P(idx), P(bp), P(x), M, D, Kp, Kp, D, P(x16), P(cp), P(st),
                                                   None, None, 0.0, P(shift), None, None, 0, 0, 0.0, 0, stream()))
            ref = x0.clone()
            lib.check(L.must3r_hip_op_gemm_sp(lib.EPI_RESID_F32, P(a), P(_split_w(Wp)), P(vals), P(idx), P(bp), P(ref), M, D, Kp, Kp, D, None, None, 0, 0, stream()))
        else:
            lib.check(L.must3r_hip_op_gemm_fold256(lib.EPI_RESID_F32, 0, P(a), P(Wp.half()), None, None, P(bp), P(x), M, D, Kp, Kp, D, P(x16), P(cp), P(st),
                                                   None, None, 0.0, P(shift), None, None, 0, 0, 0.0, 0, stream()))
            ref = x0.clone()
            lib.check(L.must3r_hip_op_gemm(1, lib.EPI_RESID_F32, P(a), P(Wp.half()), P(bp), P(ref), M, D, Kp, Kp, D, None, None, 0, 0, None, 0, 0, 0, 0, 0, 0, 0, stream()))
        torch.cuda.synchronize()
        # the fp32 rows are those of the unfolded launch, bit for bit; the copy, the shifted fp16 rows and the wave-tile sums follow from them
        assert torch.equal(x, ref) and torch.equal(cp, x), prod
        ysh = x - shift0[:, None]
        assert torch.equal(x16, ysh.half()), prod
        fr = ysh.double().view(M, D // 64, 64)
        assert torch.allclose(st[..., 0].double(), fr.sum(-1), rtol=1e-5, atol=2e-4), prod
        assert torch.allclose(st[..., 1].double(), (fr * fr).sum(-1), rtol=1e-5, atol=2e-3), prod
        assert torch.equal(shift, shift0), "a producer reads the shift, it does not write it"

        # ---- consumers of these rows
        gam = 1.0 + 0.3 * torch.randn((D,), device="cuda", generator=g)
        bet = 0.2 * torch.randn((D,), device="cuda", generator=g)
        ln = torch.nn.functional.layer_norm(x.double(), (D,), gam.double(), bet.double(), 1e-6)
        h16 = torch.empty((M, D), device="cuda", dtype=torch.float16)
        lib.check(L.must3r_hip_op_layernorm(1, P(x), None, P(gam), P(bet), P(h16), None, None, None, M, D, 1e-6, stream()))
        gh, gw = 24, 32
        ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
        pos = torch.stack((ys.reshape(-1), xs.reshape(-1)), -1).repeat(M // (gh * gw), 1).contiguous().cuda()
        buf = (C.c_float * (64 * 32))()
        L.must3r_hip_rope_table(100.0, 1.0, 64, buf)
        tab = torch.tensor(list(buf), device="cuda")
        cons = (("projq", lib.EPI_STORE16, D, 0.18, 2), ("qkv", lib.EPI_QKV_ROPE, 3 * D, 0.18, 2)) if split else (("fc1", lib.EPI_STORE16_GELU, 4 * D, 0.0, 0),)
        for name, epi, N, scale, ws in cons:
            W = torch.randn((N, D), device="cuda", generator=g) / math.sqrt(D)
            b = torch.randn((N,), device="cuda", generator=g)
            Wg = W * gam
            s_n = Wg.double().sum(1).float()
            c_n = (W.double() @ bet.double() + b.double()).float()
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
            plain = torch.empty((M, N), device="cuda", dtype=torch.float16)
            rope = epi == lib.EPI_QKV_ROPE
            sh = shift0.clone()
            rp = (P(pos), P(tab), 2 * D, 64) if rope else (None, None, 0, 0)
            if ws == 2:
                vals, idx = sparse(Wg)
                lib.check(L.must3r_hip_op_gemm_fold256(epi, 2, P(x16), P(_split_w(Wg)), P(vals), P(idx), P(c_n), P(out), M, N, D, D, N, None, None, None, P(st), P(s_n),
                                                       1e-6, P(sh), *rp, scale, D if scale else 0, stream()))
                vals, idx = sparse(W)
                lib.check(L.must3r_hip_op_gemm_sp(epi, P(h16), P(_split_w(W)), P(vals), P(idx), P(b), P(plain), M, N, D, D, N, *rp, stream()))
            else:
                lib.check(L.must3r_hip_op_gemm_fold256(epi, 0, P(x16), P(Wg.half()), None, None, P(c_n), P(out), M, N, D, D, N, None, None, None, P(st), P(s_n),
                                                       1e-6, P(sh), *rp, scale, D if scale else 0, stream()))
                lib.check(L.must3r_hip_op_gemm(1, epi, P(h16), P(W.half()), P(b), P(plain), M, N, D, D, N, None, None, 0, 0, None, 0, 0, 0, 0, 0, 0, 0, stream()))
            torch.cuda.synchronize()
            y = ln[:2048] @ W.double().t() + b.double()   # (fp64 reference on the first 2048 rows: they hold the stress rows)
            if epi == lib.EPI_STORE16_GELU:
                y = torch.nn.functional.gelu(y)
            if rope:
                yy = y.cpu().view(1, 2048, 3, heads, 64)
                q, k, v = (yy[:, :, i].permute(0, 2, 1, 3).float() for i in range(3))
                pp = pos[:2048].cpu().view(1, 2048, 2)
                y = torch.stack((R.rope2d(q, pp), R.rope2d(k, pp), v), dim=2).permute(0, 3, 2, 1, 4).reshape(2048, N).double().cuda()
            y_plain = y
            if scale:
                y = y.clone()
                y[:, :D] *= scale
            e = (rel_inf(out[:2048], y), rel_inf(plain[:2048], y_plain))
            errs[f"{prod}->{name}"] = e
            assert torch.isfinite(out.float()).all(), name
            # all rows: the folded and the unfolded route agree to the 16-bit rounding of the outputs
            po = plain.float().clone()
            if scale:
                po[:, :D] *= scale
            e_all = rel_inf(out, po.double())
            assert e_all < 4e-3, (name, e_all)
            # every consumer's column-0 blocks leave the rows' current mean (shift + the mean it measured) for the next producer
            assert torch.allclose(sh.double(), x.double().mean(1), rtol=1e-5, atol=2e-4), name
            assert e[0] < max(1e-3, 2 * e[1]), (name, e)
    record("gemm_fold256", D=D, errs=errs)


def test_model_fold256_switch_stays_inside_the_tolerance():
    """LNFOLD256 (an A/B instrument, off by default: measured equal in time, profiles/r06_lnfold256_ab.txt) on / off on the launches that take it: the encoder on 20 views of 384x512 (15360 rows) and a render of 28 views against a 3-view memory
    (MUST3R_512 widths, full encoder depth).  The fold changes where the rounding to fp16 happens (x - mean instead of LN(x)), nothing else: both routes sit inside the
    precision mode's tolerance of each other (measured 7.6e-4 / 7.4e-4: two independent realisations of the activation rounding, each ~7e-4 from the oracle -- the same
    picture as the sparse / dense low parts, test_sparse_low_part_batch_dependence_is_bounded); the default (off) is what the fixtures are asserted on."""
    from must3r_amd import _lib
    from must3r_amd.config import MUST3R_512
    cfg = MUST3R_512
    H, W, V = 384, 512, 20
    enc, dec = build(cfg, "fp16wa")
    imgs, ts = S.make_images(V, H, W, 3)
    imgs_c, ts_c = imgs.cuda(), ts.cuda()
    out = {}
    try:
        for mode in (1, 0, 1):
            _lib.set_option("LNFOLD256", mode)
            x, pos = enc(imgs_c, ts_c)
            mem, _ = dec(x[:3].unsqueeze(0), pos[:3].unsqueeze(0), ts[:3].unsqueeze(0), None)
            # (28 views = 21504 rows: the decoder's 768-column launches fill 252 of 256 CUs with 256 x 256 tiles -- the benched step's shape; 20 views would not)
            x28, p28, t28 = torch.cat((x, x[:8])), torch.cat((pos, pos[:8])), torch.cat((ts, ts[:8]))
            _, ren = dec(x28.unsqueeze(0), p28.unsqueeze(0), t28.unsqueeze(0), mem, render=True)
            torch.cuda.synchronize()
            out.setdefault(mode, []).append((x.clone(), ren.clone()))
    finally:
        _lib.set_option("LNFOLD256", 0)
    (x1, r1), (x1b, r1b) = out[1]
    (x0, r0), = out[0]
    assert torch.equal(x1, x1b) and torch.equal(r1, r1b), "the folded route is deterministic"
    e_x = max(rel_inf(x1[v].cpu(), x0[v].cpu()) for v in range(V))
    e_r = max(rel_inf(r1[0, v].cpu(), r0[0, v].cpu()) for v in range(V + 8))
    record("fold256_switch", encoder_tokens=e_x, render_pointmaps=e_r)
    assert e_x > 0.0 and e_r > 0.0, "the fold did not run: the test is vacuous"
    assert e_x < TOL["fp16wa"] and e_r < TOL["fp16wa"], (e_x, e_r)


def test_default_library_refuses_the_parked_fp8_attention_flag():
    """r06: the e4m3 attention path (BASELINE.json configs[4]; +0.4 % at 1.2e-3 ... 1.4e-3 from the 16-bit path) is compiled only with -DM3R_ATTN_FP8.  The default
    library says so instead of computing: the module property raises, the C entry points return status 1 with an error string that names the build flag."""
    import ctypes as C
    from must3r_amd import _lib
    if _lib.has_fp8_attention():
        pytest.skip("experiment build: the flag is honoured (tests/test_model_gpu.py::test_fp8_attention_*)")
    enc, dec = build(SMALL, "fp16wa")
    with pytest.raises(RuntimeError, match="M3R_ATTN_FP8"):
        enc.attention_fp8 = True
    assert enc.attention_fp8 is False
    L = _lib.load()
    q = torch.zeros((64, 64), device="cuda", dtype=torch.float16)
    tab = torch.tensor([(0, 64, 0, 64, 0, 0)], dtype=torch.int32, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    rc = L.must3r_hip_op_attention(_lib.F16 | _lib.ATTN_FP8, P(q), P(q), P(q), P(q), 64, 64, 64, 64, 1, P(tab), 1, 64, 0, None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"M3R_ATTN_FP8" in L.must3r_hip_last_error()


# ------------------------------------------------------------------------------------------------------------------------------------------------
# r06: CausalMUSt3R.forward (must3r/model/decoder.py:352-553; SURVEY.md section 8f "later")
# ------------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["fp16w2", "fp16wa"])
def test_causal_forward_matches_the_reference_fixture(precision):
    """The reference's CausalMUSt3R (memory dropout off) on the leaf shims wrote tests/golden/small_224_causal.npz (oracle/make_golden.py main_causal): calls of
    [3, 2, 1] views -- the first one against an EMPTY memory, where view 0 attends view 1's tokens (decoder.py:399-402) and views 1, 2 the views before them -- then a
    render of all six.  The HIP module (must3r_hip_decode_args.causal: a key PREFIX per view) must give the same pointmaps, memory, labels and tuple tails, and something
    else than MUSt3R's own-token rule on the same inputs."""
    import numpy as np
    import must3r_amd.model as M
    from util import rel_inf_view
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "small_224_causal.npz"))
    H, W, Vn, ps, tks = (int(v) for v in fx["meta"][:5])
    calls = [int(v) for v in fx["meta"][5:]]
    cfg = SMALL
    enc, plain = build(cfg, precision)
    dec = M.CausalMUSt3R(img_size=(cfg.img_size,) * 2, enc_embed_dim=cfg.enc_dim, embed_dim=cfg.dec_dim, depth=cfg.dec_depth, num_heads=cfg.dec_heads,
                         feedback_type="single_mlp", memory_mode="kv", landscape_only=False, use_mem_mask=True)
    dec.load_state_dict(S.make_decoder_state_dict(cfg, 0), strict=True)
    dec = dec.cuda().eval()
    dec.precision = precision
    imgs, ts = S.make_images(Vn, H, W, 0)
    x, pos = enc(imgs.cuda(), ts)
    mem, upd, i = None, [], 0
    for k, nb in enumerate(calls):
        a = (x[i:i + nb].unsqueeze(0), pos[i:i + nb].unsqueeze(0), ts[i:i + nb].unsqueeze(0))
        if i == 0:
            _, pm_plain = plain(*a, None)
        mem, pm = dec(*a, mem)
        upd.append(pm[0])
        assert [int(v) for v in mem[2:]] == [int(v) for v in fx["tails"][k]], (k, mem[2:])
        i += nb
    _, ren = dec(x.unsqueeze(0), pos.unsqueeze(0), ts.unsqueeze(0), mem, render=True)
    torch.cuda.synchronize()
    upd = torch.cat(upd, 0).cpu()
    e_u = [rel_inf_view(upd[v, ::ps, ::ps], fx["update"][v], fx["update_vmax"][v]) for v in range(Vn)]
    e_r = [rel_inf_view(ren[0, v, ::ps, ::ps].cpu(), fx["render"][v], fx["render_vmax"][v]) for v in range(Vn)]
    e_m = rel_inf(mem[0][-1][0, ::tks, ::tks].float().cpu(), torch.from_numpy(fx["mem_last"]))
    record("causal_forward", precision=precision, update_per_view=e_u, render_per_view=e_r, mem_last=e_m)
    assert np.array_equal(mem[1].cpu().numpy(), fx["labels"])
    assert max(e_u) < TOL[precision] and max(e_r) < TOL[precision] and e_m < 2 * TOL[precision], (e_u, e_r, e_m)
    # the causal rule is a different computation: MUSt3R's init call lets view 0 attend views 1 AND 2, view 1 attend views 0 AND 2, ...
    assert rel_inf(pm_plain[0].cpu(), upd[:calls[0]]) > 1e-2
    with pytest.raises(TypeError):
        dec([x[:1].unsqueeze(0)], [pos[:1].unsqueeze(0)], [ts[:1].unsqueeze(0)], mem)


def test_causal_forward_batched_scenes_equal_their_single_scene_calls():
    """B = 2 scenes through ONE causal call (the batch dimension of decoder.py:437: the scenes never interact): the key prefixes are per scene, so every scene must come
    out as from its own call -- init call of 3 views, update of 2, render of 5."""
    import must3r_amd.model as M
    cfg = TINY
    enc, _ = build(cfg, "fp16w2")
    dec = M.CausalMUSt3R(img_size=(cfg.img_size,) * 2, enc_embed_dim=cfg.enc_dim, embed_dim=cfg.dec_dim, depth=cfg.dec_depth, num_heads=cfg.dec_heads,
                         feedback_type="single_mlp", memory_mode="kv", landscape_only=False)
    dec.load_state_dict(S.make_decoder_state_dict(cfg, 0), strict=True)
    dec = dec.cuda().eval()
    dec.precision = "fp16w2"
    xs, ps = [], []
    for b in range(2):
        imgs, ts = S.make_images(5, 48, 64, 20 + b)
        x, pos = enc(imgs.cuda(), ts)
        xs.append(x); ps.append(pos)
    X, P_ = torch.stack(xs), torch.stack(ps)
    T = ts.unsqueeze(0).expand(2, -1, -1)
    mem, pm0 = dec(X[:, :3], P_[:, :3], T[:, :3], None)
    mem, pm1 = dec(X[:, 3:5], P_[:, 3:5], T[:, 3:5], mem)
    _, ren = dec(X, P_, T, mem, render=True)
    for b in range(2):
        m, q0 = dec(X[b:b + 1, :3], P_[b:b + 1, :3], T[b:b + 1, :3], None)
        m, q1 = dec(X[b:b + 1, 3:5], P_[b:b + 1, 3:5], T[b:b + 1, 3:5], m)
        _, qr = dec(X[b:b + 1], P_[b:b + 1], T[b:b + 1], m, render=True)
        torch.cuda.synchronize()
        e = max(rel_inf(pm0[b].cpu(), q0[0].cpu()), rel_inf(pm1[b].cpu(), q1[0].cpu()), rel_inf(ren[b].cpu(), qr[0].cpu()))
        assert e < 0.25 * TOL["fp16w2"], (b, e)
        assert tuple(int(v) for v in mem[2:]) == tuple(int(v) for v in m[2:]) == (5, 1, 12)
