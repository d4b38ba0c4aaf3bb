"""B > 1 through the reference's decoder API (decoder.py:158-350 takes [B, nimgs, N, C]).  Batch elements are independent
in the reference (pinned by tests/test_oracle_vs_reference.py::test_batch_elements_are_independent_in_reference), so the
oracle for a batch is the B = 1 oracle per element.  The HIP route decodes the B scenes with ONE native call
(must3r_hip_decode_args.n_scenes: M = B x rows in every GEMM, per-scene rows of the [B, capacity, mem_D] memory buffers); every
GEMM tile shape accumulates in the same order, so where no launch changes its ALGORITHM with the batch (split-KV factor, LN
fold) the batched results are bit-identical to the per-scene calls, elsewhere within the mode's tolerance.
(File named to run last: it is the newest test of the round.)"""
import ctypes as C

import pytest
import torch

from must3r_amd import synthetic as S
from must3r_amd.config import TINY, SMALL, MUST3R_512
from util import TOL, rel_inf
from test_model_gpu import build
from test_ops_gpu import record

pytestmark = pytest.mark.gpu


def _c(*a):
    return [v.contiguous() for v in a]


def test_batched_decode_equals_per_scene_and_oracle():
    from oracle import must3r_ref as R
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(6, 48, 64, 9)
    x, pos = enc(imgs.cuda(), ts.cuda())
    x, pos, t = x.view(2, 3, *x.shape[1:]), pos.view(2, 3, *pos.shape[1:]), ts.cuda().view(2, 3, 2)
    c = _c
    mem, pm_a, feats = dec(*c(x[:, :2], pos[:, :2], t[:, :2]), None, return_feats=True)
    mem, pm_b = dec(*c(x[:, 2:], pos[:, 2:], t[:, 2:]), mem)
    mem_r, pm_r = dec(x, pos, t, mem, render=True)
    assert pm_a.shape == (2, 2, 48, 64, 7) and pm_b.shape == (2, 1, 48, 64, 7) and pm_r.shape == (2, 3, 48, 64, 7)
    assert mem[0][0].shape[:2] == (2, 36) and mem[1].shape == (2, 36) and tuple(mem[2:]) == (3, 3, 36)
    assert len(feats) == cfg.dec_depth + 1 and feats[0].shape[:3] == (2, 2, 12) and feats[-1].shape == (2, 2, 12, cfg.dec_dim)
    assert mem_r is mem
    # list form
    mem_l, pm_l = dec.forward_list([x[:, :2].contiguous()], [pos[:, :2].contiguous()], [t[:, :2].contiguous()], None)
    assert torch.equal(pm_l[0], pm_a)

    xo, po = R.encoder_forward(sde, cfg, imgs, ts)
    xo, po, to = xo.view(2, 3, *xo.shape[1:]), po.view(2, 3, *po.shape[1:]), ts.view(2, 3, 2)
    for b in range(2):
        s = slice(b, b + 1)
        m1, p1a, f1 = dec(*c(x[s, :2], pos[s, :2], t[s, :2]), None, return_feats=True)
        m1, p1b = dec(*c(x[s, 2:], pos[s, 2:], t[s, 2:]), m1)
        _, p1r = dec(*c(x[s], pos[s], t[s]), m1, render=True)
        assert torch.equal(pm_a[s], p1a) and torch.equal(pm_b[s], p1b) and torch.equal(pm_r[s], p1r)
        assert all(torch.equal(a[s], v) for a, v in zip(mem[0], m1[0])) and torch.equal(mem[1][s], m1[1])
        assert all(torch.equal(a[s], v) for a, v in zip(feats, f1))
        mo, oa = R.decoder_forward(sdd, cfg, *c(xo[s, :2], po[s, :2], to[s, :2]), None, False, "kv")
        mo, ob = R.decoder_forward(sdd, cfg, *c(xo[s, 2:], po[s, 2:], to[s, 2:]), mo, False, "kv")
        _, orr = R.decoder_forward(sdd, cfg, *c(xo[s], po[s], to[s]), mo, True, "kv")
        for got, want in ((pm_a[s], oa), (pm_b[s], ob), (pm_r[s], orr)):
            assert rel_inf(got.cpu(), torch.as_tensor(want)) < TOL["fp16w2"]


@pytest.mark.parametrize("precision", ["fp16w2", "bf16"])
def test_scenes_in_flight_match_oracle_and_single_scene_runs(precision):
    """engine.run_scenes (S scenes ride the decoder's batch dimension, one native call per schedule step) against the oracle of
    EVERY scene and against engine.run_scene of each scene alone."""
    from must3r_amd.engine import run_scene, run_scenes
    from oracle import must3r_ref as R
    cfg = SMALL
    enc, dec = build(cfg, precision)
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    Sn, V, H, W = 3, 4, 224, 224
    imgs = torch.stack([S.make_images(V, H, W, 30 + b)[0] for b in range(Sn)])
    ts = S.make_images(V, H, W, 0)[1]
    out = run_scenes(enc, dec, imgs.cuda(), ts)
    torch.cuda.synchronize()
    assert out["update"].shape == (Sn, V, H, W, 7) and out["render"].shape == (Sn, V, H, W, 7)
    assert out["mem"][0][0].shape[:2] == (Sn, V * 196) and out["mem"][1].shape == (Sn, V * 196)
    assert out["pts3d"].shape == (Sn, V, H, W, 3) and out["conf"].shape == (Sn, V, H, W)
    errs, diffs = [], []
    for b in range(Sn):
        with torch.no_grad():
            upd_o, ren_o, mem_o = R.run_scene(sde, sdd, cfg, imgs[b], ts)
        errs.append(max(rel_inf(out["update"][b].cpu(), upd_o), rel_inf(out["render"][b].cpu(), ren_o)))
        one = run_scene(enc, dec, imgs[b].cuda(), ts)
        diffs.append(max(rel_inf(out["update"][b].cpu(), one["update"].cpu()), rel_inf(out["render"][b].cpu(), one["render"].cpu())))
        assert torch.equal(out["mem"][1][b:b + 1].cpu(), one["mem"][1].cpu()) and tuple(out["mem"][2:]) == tuple(one["mem"][2:])
    record("scenes_in_flight", precision=precision, vs_oracle=errs, vs_single=diffs)
    assert max(errs) < TOL[precision], errs
    assert max(diffs) < TOL[precision], diffs


@pytest.mark.parametrize("mode", ["norm_y", "raw"])
def test_batched_decode_memory_modes(mode):
    """memory_mode 'norm_y' / 'raw' (layers.py:81-88) with B = 2: per-scene token memories, K|V projected per call."""
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    dec.change_memory_mode(mode)
    try:
        imgs, ts = S.make_images(6, 48, 64, 5)
        x, pos = enc(imgs.cuda(), ts.cuda())
        x, pos, t = x.view(2, 3, *x.shape[1:]), pos.view(2, 3, *pos.shape[1:]), ts.view(2, 3, 2)
        mem, pa = dec(*_c(x[:, :2], pos[:, :2], t[:, :2]), None)
        mem, pb = dec(*_c(x[:, 2:], pos[:, 2:], t[:, 2:]), mem)
        _, pr = dec(x, pos, t, mem, render=True)
        assert mem[0][0].shape == (2, 36, cfg.dec_dim)
        for b in range(2):
            s = slice(b, b + 1)
            m1, qa = dec(*_c(x[s, :2], pos[s, :2], t[s, :2]), None)
            m1, qb = dec(*_c(x[s, 2:], pos[s, 2:], t[s, 2:]), m1)
            _, qr = dec(*_c(x[s], pos[s], t[s]), m1, render=True)
            assert torch.equal(pa[s], qa) and torch.equal(pb[s], qb) and torch.equal(pr[s], qr)
            assert all(torch.equal(a[s], v) for a, v in zip(mem[0], m1[0]))
    finally:
        dec.change_memory_mode("kv")


def test_batched_forward_list_mixed_aspect_ratios():
    """forward_list with B = 2 and two aspect ratios in one call: rows are scene-major, each scene's K|V rows keep the x_cat
    order (group, view) of decoder.py:211-214."""
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    ia, ta = S.make_images(4, 48, 64, 1)
    ib, tb = S.make_images(2, 32, 64, 2)
    xa, pa = enc(ia.cuda(), ta)
    xb, pb = enc(ib.cuda(), tb)
    xa, pa, ta = xa.view(2, 2, *xa.shape[1:]), pa.view(2, 2, *pa.shape[1:]), ta.view(2, 2, 2)
    xb, pb, tb = xb.view(2, 1, *xb.shape[1:]), pb.view(2, 1, *pb.shape[1:]), tb.view(2, 1, 2)
    mem, pms = dec([xa, xb], [pa, pb], [ta, tb], None)
    mem2, pms2 = dec([xb, xa], [pb, pa], [tb, ta], mem)
    _, prs = dec([xa, xb], [pa, pb], [ta, tb], mem2, render=True)
    assert pms[0].shape == (2, 2, 48, 64, 7) and pms[1].shape == (2, 1, 32, 64, 7)
    assert mem2[0][0].shape[:2] == (2, 2 * (2 * 12 + 8))
    for b in range(2):
        s = slice(b, b + 1)
        m1, q = dec(_c(xa[s], xb[s]), _c(pa[s], pb[s]), _c(ta[s], tb[s]), None)
        m2, q2 = dec(_c(xb[s], xa[s]), _c(pb[s], pa[s]), _c(tb[s], ta[s]), m1)
        _, qr = dec(_c(xa[s], xb[s]), _c(pa[s], pb[s]), _c(ta[s], tb[s]), m2, render=True)
        for got, want in zip(pms + pms2 + prs, q + q2 + qr):
            assert torch.equal(got[s], want)
        assert all(torch.equal(a[s], v) for a, v in zip(mem2[0], m2[0])) and torch.equal(mem2[1][s], m2[1])


def test_batched_decode_fp8_attention():
    """MUST3R_ATTN_FP8 with B = 2: [K e4m3 | V fp16] memory rows per scene (grouped quantisation into each scene's rows)."""
    from must3r_amd import _lib as _l
    if not _l.has_fp8_attention():
        pytest.skip("the e4m3 attention path is parked: built only with make EXTRA=-DM3R_ATTN_FP8 (include/must3r_hip.h)")
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    enc.attention_fp8 = dec.attention_fp8 = True
    try:
        imgs, ts = S.make_images(6, 48, 64, 11)
        x, pos = enc(imgs.cuda(), ts)
        x, pos, t = x.view(2, 3, *x.shape[1:]), pos.view(2, 3, *pos.shape[1:]), ts.view(2, 3, 2)
        mem, pa = dec(*_c(x[:, :2], pos[:, :2], t[:, :2]), None)
        mem, pb = dec(*_c(x[:, 2:], pos[:, 2:], t[:, 2:]), mem)
        _, pr = dec(x, pos, t, mem, render=True)
        assert mem[0][0].dtype == torch.uint8 and mem[0][0].shape[2] == 3 * cfg.dec_dim   # rows [K e4m3 | V fp16]
        for b in range(2):
            s = slice(b, b + 1)
            m1, qa = dec(*_c(x[s, :2], pos[s, :2], t[s, :2]), None)
            m1, qb = dec(*_c(x[s, 2:], pos[s, 2:], t[s, 2:]), m1)
            _, qr = dec(*_c(x[s], pos[s], t[s]), m1, render=True)
            assert torch.equal(pa[s], qa) and torch.equal(pb[s], qb) and torch.equal(pr[s], qr)
            assert all(torch.equal(a[s], v) for a, v in zip(mem[0], m1[0]))
    finally:
        enc.attention_fp8 = dec.attention_fp8 = False


def test_render_of_more_views_than_one_table_holds_is_cut_inside_the_library():
    """The per-call view tables travel through one 64 KiB staging slot (1365 views).  A render call with more views -- the
    reference renders a whole collection in one call without max_bs, engine/inference.py:489-522 -- is cut into ranges of views
    (one scene) or scenes (B > 1) INSIDE must3r_hip_decode; an update call of that size is refused."""
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    imgs, ts = S.make_images(3, 32, 32, 3)
    x, pos = enc(imgs.cuda(), ts)
    mem, _ = dec(x[None, :2], pos[None, :2], ts[None, :2], None)
    V = 1500
    idx = torch.arange(V) % 3
    xr, pr, tr = x[idx][None].contiguous(), pos[idx][None].contiguous(), ts[idx][None]
    _, big = dec(xr, pr, tr, mem, render=True)
    _, ref = dec(x[None], pos[None], ts[None], mem, render=True)
    assert big.shape == (1, V, 32, 32, 7)
    assert torch.equal(big[0], ref[0][idx])
    # B = 2 x 750 views: cut into scenes
    mem2, _ = dec(x[None, :2].expand(2, -1, -1, -1).contiguous(), pos[None, :2].expand(2, -1, -1, -1).contiguous(), ts[None, :2].expand(2, -1, -1), None)
    _, big2 = dec(xr[:, :750].expand(2, -1, -1, -1).contiguous(), pr[:, :750].expand(2, -1, -1, -1).contiguous(), tr[:, :750].expand(2, -1, -1),
                  mem2, render=True)
    assert torch.equal(big2[0], big[0, :750]) and torch.equal(big2[1], big[0, :750])
    with pytest.raises(RuntimeError, match="views in one memory update"):
        dec(xr, pr, tr, mem)


def test_decode_refuses_a_call_that_would_overrun_the_memory_buffers():
    """ABI 5: must3r_hip_decode_args.mem_capacity is checked against n_mem + the rows the call appends."""
    from must3r_amd import _lib
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    imgs, ts = S.make_images(2, 32, 32, 3)
    x, pos = enc(imgs.cuda(), ts)
    ctx = dec._context()
    N, D = 4, cfg.dec_dim
    bufs = [torch.zeros((1, 6, 2 * D), dtype=torch.float16, device="cuda") for _ in range(cfg.dec_depth)]   # 6 rows < 2 views x 4 tokens
    pm = torch.empty((1, 2, 32, 32, 7), device="cuda")
    groups = (_lib.Group * 1)(_lib.Group(x.data_ptr(), pos.data_ptr(), 2, N, 32, 32, pm.data_ptr()))
    ptrs = (C.c_void_p * cfg.dec_depth)(*[b.data_ptr() for b in bufs])
    stream = torch.cuda.current_stream().cuda_stream
    for cap, frag in ((6, "memory buffers hold 6 rows"), (0, "mem_capacity must be given")):
        args = _lib.DecodeArgs(_lib.F16_W2, _lib.MEM_KV, 0, 1, 1, groups, 0, ptrs, None, cap, 1, 0)
        assert ctx.lib.must3r_hip_decode(ctx.handle, C.byref(args), stream) != 0
        assert frag in ctx.lib.must3r_hip_last_error().decode()
    args = _lib.DecodeArgs(_lib.F16_W2, _lib.MEM_KV, 0, 1, 1, groups, 0, ptrs, None, 8, 2, 4)   # scene stride < capacity
    assert ctx.lib.must3r_hip_decode(ctx.handle, C.byref(args), stream) != 0
    assert "mem_scene_stride" in ctx.lib.must3r_hip_last_error().decode()
    torch.cuda.synchronize()
    assert all(float(b.abs().sum()) == 0.0 for b in bufs)    # nothing was written


def test_full_size_scenes_in_flight_properties():
    """BASELINE geometry (384x512, ViT-L / ViT-B): 2 scenes x 4 views in flight against the same scenes run alone -- the batched
    calls take other tile shapes and, for the one-view updates, another algorithm (no LN fold), so the comparison is the mode's
    tolerance; encoder tokens are bit-identical (batch invariance of the encoder)."""
    from must3r_amd.engine import run_scene, run_scenes
    cfg = MUST3R_512
    enc, dec = build(cfg, "fp16w2")
    Sn, V, H, W = 2, 4, 384, 512
    imgs = torch.stack([S.make_images(V, H, W, 50 + b)[0] for b in range(Sn)]).cuda()
    ts = S.make_images(V, H, W, 0)[1]
    out = run_scenes(enc, dec, imgs, ts)
    diffs = []
    for b in range(Sn):
        one = run_scene(enc, dec, imgs[b], ts)
        assert torch.equal(out["x"][b], one["x"])
        diffs.append(max(rel_inf(out["update"][b].cpu(), one["update"].cpu()), rel_inf(out["render"][b].cpu(), one["render"].cpu())))
    record("scenes_in_flight_full_size", diffs=diffs)
    assert torch.isfinite(out["render"]).all()
    assert max(diffs) < TOL["fp16w2"], diffs
