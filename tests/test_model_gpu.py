"""GPU (-m gpu): the two HIP forwards behind the reference's nn.Module API against
(a) the committed fixtures produced by the REAL reference (tests/golden), (b) the fp32 CPU oracle on the
same seeded inputs, and (c) size-independent properties at the BASELINE geometry (512x384, ViT-L/ViT-B)."""
import numpy as np
import pytest
import torch

from must3r_amd import synthetic as S
from must3r_amd.config import TINY, SMALL, MUST3R_224, MUST3R_512
from util import TOL, PRECISIONS, load_golden, rel_inf, rel_inf_view, rel_l2
from test_ops_gpu import record

pytestmark = pytest.mark.gpu

CASES = {"tiny_48x64_v4": TINY, "small_224_v3": SMALL, "must3r224_v2": MUST3R_224}
_models = {}


def build(cfg, precision, seed=0):
    """HIP-backed modules with the seeded synthetic weights (one pair per geometry; precision is a runtime switch)."""
    import must3r_amd.model as M
    key = (cfg, seed)
    if key not in _models:
        enc = M.Dust3rEncoder(img_size=(cfg.img_size,) * 2, embed_dim=cfg.enc_dim, depth=cfg.enc_depth, num_heads=cfg.enc_heads)
        dec = M.MUSt3R(img_size=(cfg.img_size,) * 2, enc_embed_dim=cfg.enc_dim, embed_dim=cfg.dec_dim, depth=cfg.dec_depth,
                       num_heads=cfg.dec_heads, feedback_type="single_mlp", memory_mode="kv", landscape_only=False)
        enc.load_state_dict(S.make_encoder_state_dict(cfg, seed), strict=True)
        dec.load_state_dict(S.make_decoder_state_dict(cfg, seed), strict=True)
        _models[key] = (enc.cuda().eval(), dec.cuda().eval())
    enc, dec = _models[key]
    enc.precision = dec.precision = precision
    return enc, dec


def hip_scene(cfg, precision, H, W, V, mb, seed=0):
    from must3r_amd.engine import run_scene
    enc, dec = build(cfg, precision)
    imgs, ts = S.make_images(V, H, W, seed)
    out = run_scene(enc, dec, imgs.cuda(), ts.cuda(), mem_batches=mb)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("precision", list(PRECISIONS))
@pytest.mark.parametrize("name", list(CASES))
def test_scene_matches_reference_fixture(name, precision):
    g = load_golden(name)
    H, W, V, ps, tks = (int(v) for v in g["meta"][:5])
    mb = [int(v) for v in g["meta"][5:]]
    out = hip_scene(CASES[name], precision, H, W, V, mb)
    tol = TOL[precision]
    x, upd, ren, mem = out["x"].cpu(), out["update"].cpu(), out["render"].cpu(), out["mem"]
    errs = dict(x=rel_inf(x[:, ::tks, ::tks], g["x"]), update=rel_inf(upd[:, ::ps, ::ps], g["update"]),
                render=rel_inf(ren[:, ::ps, ::ps], g["render"]), render_l2=rel_l2(ren[:, ::ps, ::ps], g["render"]),
                mem_first=rel_inf(mem[0][0][0, ::tks, ::tks].float().cpu(), g["mem_first"]),
                mem_last=rel_inf(mem[0][-1][0, ::tks, ::tks].float().cpu(), g["mem_last"]),
                render_maxabs=float((ren[:, ::ps, ::ps] - torch.from_numpy(g["render"])).abs().max()))
    record("scene_vs_fixture", case=name, precision=precision, **errs)
    assert np.array_equal(out["pos"].cpu()[:, ::tks].numpy(), g["pos"])
    assert np.array_equal(mem[1].cpu().numpy(), g["labels"]) and [int(v) for v in mem[2:]] == [int(v) for v in g["tail"]]
    assert torch.isfinite(ren).all() and torch.isfinite(upd).all()
    assert errs["x"] < tol and errs["update"] < tol and errs["render"] < tol, errs
    # memory is stored as 16-bit K|V: one extra rounding on top of the path error
    u = 2.0 ** -8 if precision == "bf16" else 2.0 ** -11
    assert mem[0][0].dtype == (torch.bfloat16 if precision == "bf16" else torch.float16)
    assert errs["mem_first"] < tol + u and errs["mem_last"] < tol + u, errs


FP8_TOL = 1.0e-2   # stated tolerance of the fp8-attention mode (pointmaps, rel-inf vs the reference); emulated 2e-3 (scripts/emul/fp8_attention.py)


@pytest.mark.parametrize("name", ["tiny_48x64_v4", "small_224_v3", "must3r224_v2", "must3r224_v10"])
def test_fp8_attention_scene_vs_reference_fixture(name):
    """BASELINE.json configs[4] "fp8 MFMA attention path": Q and K as e4m3 through the MX-scaled 32x32x64 MFMA in every chip-filling
    attention of the encoder and the decoder (memory rows [K e4m3 | V fp16]), P / V and everything else as in fp16w2.  Its error is
    reported next to fp16w2's: it does NOT meet the 1e-3 target (3 mantissa bits on Q and K), the assertion is the mode's own stated
    tolerance (1e-2)."""
    from must3r_amd import _lib as _l
    if not _l.has_fp8_attention():
        pytest.skip("the e4m3 attention path is parked: built only with make EXTRA=-DM3R_ATTN_FP8 (include/must3r_hip.h)")
    g = load_golden(name)
    cfg = dict(CASES, **BIG_CASES)[name]
    H, W, V, ps, tks = (int(v) for v in g["meta"][:5])
    mb = [int(v) for v in g["meta"][5:]]
    enc, dec = build(cfg, "fp16w2")
    base = hip_scene(cfg, "fp16w2", H, W, V, mb)
    enc.attention_fp8 = dec.attention_fp8 = True
    try:
        out = hip_scene(cfg, "fp16w2", H, W, V, mb)
    finally:
        enc.attention_fp8 = dec.attention_fp8 = False
    assert out["mem"][0][0].dtype == torch.uint8 and out["mem"][0][0].shape[2] == 3 * cfg.dec_dim and base["mem"][0][0].dtype == torch.float16
    errs = {}
    for tag, o in (("fp8", out), ("fp16w2", base)):
        errs[tag] = dict(x=rel_inf(o["x"].cpu()[:, ::tks, ::tks], g["x"]), update=rel_inf(o["update"].cpu()[:, ::ps, ::ps], g["update"]),
                         render=rel_inf(o["render"].cpu()[:, ::ps, ::ps], g["render"]),
                         render_l2=rel_l2(o["render"].cpu()[:, ::ps, ::ps], g["render"]))
    record("fp8_attention_vs_fixture", case=name, **{f"{t}_{k}": v for t, e in errs.items() for k, v in e.items()})
    assert torch.isfinite(out["render"]).all() and torch.isfinite(out["update"]).all()
    assert np.array_equal(out["mem"][1].cpu().numpy(), g["labels"])
    # (the 12-token views of the tiny fixture average the e4m3 rounding of Q / K over a few dozen keys only: 1.6e-2 measured there,
    #  5e-3 ... 1e-3 from 196 tokens per view on -- the stated tolerance is for real view sizes)
    tol8 = 3 * FP8_TOL if name.startswith("tiny") else FP8_TOL
    assert errs["fp8"]["update"] < tol8 and errs["fp8"]["render"] < tol8 and errs["fp8"]["x"] < tol8, errs
    if V * (H // 16) * (W // 16) >= 2 * 768:   # (scenes whose attention launches are all too small for the fp8 kernel stay 16-bit)
        assert errs["fp8"]["render"] > errs["fp16w2"]["render"]      # the flag really changes the arithmetic


BIG_CASES = {"must3r224_v10": MUST3R_224, "must3r512_v20": MUST3R_512}


@pytest.mark.parametrize("precision", list(PRECISIONS))
@pytest.mark.parametrize("name", list(BIG_CASES))
def test_full_depth_scene_matches_reference_fixture(name, precision):
    """Full-depth parity at the benchmark configurations (BASELINE.json configs[1] / configs[2]) against outputs of the
    REAL reference (oracle/make_golden.py model_big): every view of the sequential memory update [2,1,...,1]
    (engine/inference.py:396-442) and of the render pass against the final memory (:489-522), per view, plus the first and
    last layer of the final memory -- error growth with memory depth at full width is measured, not inferred."""
    g = load_golden(name)
    H, W, V, ps, tks = (int(v) for v in g["meta"][:5])
    mb = [int(v) for v in g["meta"][5:]]
    out = hip_scene(BIG_CASES[name], precision, H, W, V, mb)
    tol = TOL[precision]
    x, upd, ren, mem = out["x"].cpu(), out["update"].cpu(), out["render"].cpu(), out["mem"]
    assert upd.shape[0] == V and ren.shape[0] == V and mem[0][0].shape[1] == V * (H // 16) * (W // 16)
    # per view: the error over the stored (sub-sampled) pixels against that view's FULL-resolution range (r05: `*_vmax` of the fixture, the normalisation of
    # bench.py's all-pixel parity figure; the max of the SAMPLE under-states a view's range by up to 40 % and was the only reason a view ever reached 1e-3)
    upd_v = [rel_inf_view(upd[v, ::ps, ::ps], g["update"][v], g["update_vmax"][v]) for v in range(V)]
    ren_v = [rel_inf_view(ren[v, ::ps, ::ps], g["render"][v], g["render_vmax"][v]) for v in range(V)]
    errs = dict(x=rel_inf(x[:, ::tks, ::tks], g["x"]), update=rel_inf(upd[:, ::ps, ::ps], g["update"]),
                render=rel_inf(ren[:, ::ps, ::ps], g["render"]), render_l2=rel_l2(ren[:, ::ps, ::ps], g["render"]),
                update_view_max=max(upd_v), render_view_max=max(ren_v), update_last_view=upd_v[-1],
                mem_first=rel_inf(mem[0][0][0, ::tks, ::tks].float().cpu(), g["mem_first"]),
                mem_last=rel_inf(mem[0][-1][0, ::tks, ::tks].float().cpu(), g["mem_last"]))
    record("full_depth_vs_fixture", case=name, precision=precision, views=V, update_per_view=[round(e, 6) for e in upd_v],
           render_per_view=[round(e, 6) for e in ren_v], **errs)
    assert np.array_equal(mem[1].cpu().numpy(), g["labels"]) and [int(v) for v in mem[2:]] == [int(v) for v in g["tail"]]
    assert torch.isfinite(ren).all() and torch.isfinite(upd).all()
    assert errs["x"] < tol and errs["update"] < tol and errs["render"] < tol, errs
    assert max(upd_v) < tol and max(ren_v) < tol, (upd_v, ren_v)          # every single view, normalised by its own range
    # r06: ALL pixels of the views that are worst against the oracle (oracle/make_golden.py full_views: the real reference's full-resolution maps of
    # update views 1 / 15 and render views 2 / 19 of the headline scene) -- the all-pixel figure is a figure against the REFERENCE here, not against the port
    import os
    from conftest import GOLDEN
    if os.path.exists(os.path.join(GOLDEN, name + "_fullviews.npz")):
        gf = load_golden(name + "_fullviews")
        full_u = [rel_inf(upd[int(v)], gf["update"][k]) for k, v in enumerate(gf["update_views"])]
        full_r = [rel_inf(ren[int(v)], gf["render"][k]) for k, v in enumerate(gf["render_views"])]
        record("full_depth_vs_fixture_all_pixels", case=name, precision=precision, update_views=gf["update_views"].tolist(), render_views=gf["render_views"].tolist(),
               update=[round(e, 7) for e in full_u], render=[round(e, 7) for e in full_r])
        assert max(full_u + full_r) < tol, (full_u, full_r)
    u = 2.0 ** -8 if precision == "bf16" else 2.0 ** -11
    assert errs["mem_first"] < tol + u and errs["mem_last"] < tol + u, errs


@pytest.mark.parametrize("precision", list(PRECISIONS))
def test_mixed_aspect_ratio_list_path(precision):
    g = load_golden("tiny_mixed_ar")
    enc, dec = build(TINY, precision)
    ia, ta = S.make_images(2, 48, 64, 1)
    ib, tb = S.make_images(1, 32, 64, 2)
    L = lambda *t: [v.unsqueeze(0) for v in t]  # noqa: E731
    xa, pa = enc(ia.cuda(), ta.cuda())
    xb, pb = enc(ib.cuda(), tb.cuda())
    ta, tb = ta.cuda(), tb.cuda()
    mem, pm0 = dec(L(xa, xb), L(pa, pb), L(ta, tb), None)
    mem2, pm1 = dec(L(xb, xa), L(pb, pa), L(tb, ta), mem)
    mem3, pm2 = dec(L(xb, xa), L(pb, pa), L(tb, ta), mem2, render=True)
    assert mem3[0] is mem2[0] and mem3[2:] == mem2[2:]          # render returns the memory untouched (decoder.py:339)
    tol = TOL[precision]
    errs = {}
    for got, key in ((pm0[0], "init_a"), (pm0[1], "init_b"), (pm1[0], "upd_b"), (pm1[1], "upd_a"), (pm2[0], "ren_b"), (pm2[1], "ren_a")):
        errs[key] = rel_inf(got[0].cpu(), g[key])
    record("mixed_ar", precision=precision, **errs)
    assert max(errs.values()) < tol, errs
    assert np.array_equal(mem2[1].cpu().numpy(), g["labels"]) and [int(v) for v in mem2[2:]] == [int(v) for v in g["tail"]]
    assert rel_inf(mem2[0][-1][0].float().cpu(), g["mem_last"]) < tol + 2.0 ** -8


def test_forward_list_equals_forward_and_memory_cow():
    """SURVEY.md section 4 invariant 2, plus the append-in-place memory never aliases observable state."""
    enc, dec = build(TINY, "fp16w2")
    imgs, ts = S.make_images(4, 48, 64, 7)
    imgs, ts = imgs.cuda(), ts.cuda()
    x, pos = enc(imgs, ts)
    m_t, p_t = dec(x[:2].unsqueeze(0), pos[:2].unsqueeze(0), ts[:2].unsqueeze(0), None)
    m_l, p_l = dec([x[:2].unsqueeze(0)], [pos[:2].unsqueeze(0)], [ts[:2].unsqueeze(0)], None)
    assert torch.equal(p_t, p_l[0]) and all(torch.equal(a, b) for a, b in zip(m_t[0], m_l[0]))
    snap = [v.clone() for v in m_t[0]]
    # branch twice from the same memory (SLAM pattern: non-keyframe results are discarded, slam/model.py:520-521)
    m_a, p_a = dec(x[2:3].unsqueeze(0), pos[2:3].unsqueeze(0), ts[2:3].unsqueeze(0), m_t)
    keep_a = [v.clone() for v in m_a[0]]
    m_b, p_b = dec(x[3:4].unsqueeze(0), pos[3:4].unsqueeze(0), ts[3:4].unsqueeze(0), m_t)
    assert all(torch.equal(a, b) for a, b in zip(m_t[0], snap)), "old memory view changed"
    assert all(torch.equal(a, b) for a, b in zip(m_a[0], keep_a)), "first branch was clobbered by the second"
    m_a2, p_a2 = dec(x[2:3].unsqueeze(0), pos[2:3].unsqueeze(0), ts[2:3].unsqueeze(0), m_t)
    assert torch.equal(p_a, p_a2) and all(torch.equal(a, b) for a, b in zip(m_a[0], m_a2[0]))
    assert m_a[2:] == (3, 3, 36) and m_a[1].shape == (1, 36)


def test_caller_side_memory_surgery():
    """engine/inference.py:205-214 _remove_from_mem: the caller boolean-indexes the memory; the next call must
    accept the rebuilt tensors and equal the oracle run on the same edited memory."""
    from oracle import must3r_ref as R
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    sdd = S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(3, 48, 64, 0)
    x, pos = enc(imgs.cuda(), ts.cuda())
    tsc = ts.cuda()
    mem, _ = dec(x[:2].unsqueeze(0), pos[:2].unsqueeze(0), tsc[:2].unsqueeze(0), None)
    mem, _ = dec(x[2:].unsqueeze(0), pos[2:].unsqueeze(0), tsc[2:].unsqueeze(0), mem)
    keep = mem[1] != 1                                        # drop view 1
    vals = [v[keep].view(1, -1, v.shape[-1]) for v in mem[0]]
    labels = mem[1][keep].view(1, -1)
    edited = (vals, labels, mem[2], mem[3], mem[4])
    _, pm = dec(x.unsqueeze(0), pos.unsqueeze(0), tsc.unsqueeze(0), edited, render=True)
    mem_cpu = ([v.float().cpu() for v in vals], labels.cpu(), mem[2], mem[3], mem[4])
    _, ref = R.decoder_forward(sdd, cfg, x.cpu().unsqueeze(0), pos.cpu().unsqueeze(0), ts.unsqueeze(0), mem_cpu, True, "kv")
    e = rel_inf(pm.cpu(), ref)
    record("memory_surgery", err=e)
    assert e < TOL["fp16w2"], e
    # and an update on top of the edited memory appends after the surviving rows
    mem2, _ = dec(x[1:2].unsqueeze(0), pos[1:2].unsqueeze(0), tsc[1:2].unsqueeze(0), edited)
    assert mem2[0][0].shape[1] == 24 + 12 and torch.equal(mem2[0][1][:, :24], vals[1])


@pytest.mark.parametrize("precision", list(PRECISIONS))
def test_baseline_geometry_vs_oracle_and_properties(precision):
    """MUSt3R_512 (ViT-L enc / ViT-B dec), 384x512.  Oracle comparison on 2 views (CPU cost ~20 s) plus
    size-independent properties on 5 views: batch-invariance of the encoder and of the render pass (bit-exact),
    render does not modify the memory, labels."""
    from oracle import must3r_ref as R
    cfg = MUST3R_512
    H, W, V = 384, 512, 5
    enc, dec = build(cfg, precision)
    imgs, ts = S.make_images(V, H, W, 0)
    imgs_c, ts_c = imgs.cuda(), ts.cuda()
    x, pos = enc(imgs_c, ts_c)
    x1, pos1 = enc(imgs_c[3:4], ts_c[3:4])
    assert torch.equal(pos1[0], pos[3])
    # bit-equal in EVERY mode at this size: 5 views = 3840 rows launch no kernel that multiplies the 2:4-sparse low part (launch_epi: the RoPE qkv launch fills
    # 70 % of its rounds of 256 x 128 tiles, the N = 1024 launches have fewer than 200 tiles -- ADVICE r05), so the lone view and the batch share ONE arithmetic.
    # Where the sparse low part does run (>= 20 views in flight) its distance from the dense one is bounded by test_sparse_low_part_batch_dependence_is_bounded.
    assert torch.equal(x1[0], x[3]), "encoder is not batch-invariant"
    mem = None
    for a, b in ((0, 2), (2, 3), (3, 4), (4, 5)):
        mem, _ = dec(x[a:b].unsqueeze(0), pos[a:b].unsqueeze(0), ts_c[a:b].unsqueeze(0), mem)
    assert mem[0][0].shape == (1, V * 768, 1536) and mem[2:] == (V, V, V * 768)
    assert torch.equal(mem[1].cpu(), torch.arange(V).repeat_interleave(768).view(1, -1))
    snap = [v.clone() for v in mem[0]]
    mem_r, ren = dec(x.unsqueeze(0), pos.unsqueeze(0), ts_c.unsqueeze(0), mem, render=True)
    assert mem_r[0] is mem[0] and all(torch.equal(a, b) for a, b in zip(mem[0], snap))
    _, ren1 = dec(x[2:3].unsqueeze(0), pos[2:3].unsqueeze(0), ts_c[2:3].unsqueeze(0), mem, render=True)
    # per-view independence of the render pass (invariant 6).  Not bitwise: the split-KV factor of
    # the cross attention depends on how many views share the launch, which changes the fp32 summation order.
    e_ind = rel_inf(ren1[0, 0].cpu(), ren[0, 2].cpu())
    record("render_independence", precision=precision, err=e_ind)
    assert e_ind < 0.5 * TOL[precision], e_ind
    assert torch.isfinite(ren).all()
    # oracle on the first two views (init call + render)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    with torch.no_grad():
        xo, po = R.encoder_forward(sde, cfg, imgs[:2], ts[:2], sdpa=True)
        memo, updo = R.decoder_forward(sdd, cfg, xo.unsqueeze(0), po.unsqueeze(0), ts[:2].unsqueeze(0), None, False, "kv", sdpa=True)
        _, reno = R.decoder_forward(sdd, cfg, xo.unsqueeze(0), po.unsqueeze(0), ts[:2].unsqueeze(0), memo, True, "kv", sdpa=True)
    mem2, upd = dec(x[:2].unsqueeze(0), pos[:2].unsqueeze(0), ts_c[:2].unsqueeze(0), None)
    _, ren2 = dec(x[:2].unsqueeze(0), pos[:2].unsqueeze(0), ts_c[:2].unsqueeze(0), mem2, render=True)
    errs = dict(x=rel_inf(x[:2].cpu(), xo), update=rel_inf(upd.cpu(), updo), render=rel_inf(ren2.cpu(), reno),
                render_l2=rel_l2(ren2.cpu(), reno), render_maxabs=float((ren2.cpu() - reno).abs().max()))
    record("baseline_geometry_vs_oracle", precision=precision, **errs)
    tol = TOL[precision]
    assert errs["x"] < tol and errs["update"] < tol and errs["render"] < tol, errs


def test_autocast_selects_operand_dtype():
    enc, dec = build(TINY, "fp16")
    imgs, ts = S.make_images(2, 48, 64, 0)
    x, pos = enc(imgs.cuda(), ts.cuda())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        mem, _ = dec(x.unsqueeze(0), pos.unsqueeze(0), ts.cuda().unsqueeze(0), None)
    assert mem[0][0].dtype == torch.bfloat16       # memory dtype follows autocast like decoder.py:142 / blocks/__init__.py:5
    mem, _ = dec(x.unsqueeze(0), pos.unsqueeze(0), ts.cuda().unsqueeze(0), None)
    assert mem[0][0].dtype == torch.float16


@pytest.mark.parametrize("precision", ["fp16w2", "bf16"])
def test_memory_modes_norm_y_raw_kv(precision):
    """SURVEY.md section 4 invariant 1: 'norm_y', 'kv' and 'raw' differ only in WHAT is cached (layers.py:81-99).
    Natively: 'norm_y' stores the same 16-bit LayerNorm output the 'kv' projection consumes, so its pointmaps are
    bit-identical to 'kv'; 'raw' stores the 16-bit tokens and normalises them again at use (tolerance)."""
    from oracle import must3r_ref as R
    cfg = TINY
    enc, dec = build(cfg, precision)
    sdd = S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(4, 48, 64, 5)
    x, pos = enc(imgs.cuda(), ts.cuda())
    tsc = ts.cuda()
    outs, mems = {}, {}
    try:
        for mode in ("kv", "norm_y", "raw"):
            dec.change_memory_mode(mode)
            mem = None
            pms = []
            for a, b in ((0, 2), (2, 3), (3, 4)):
                mem, pm = dec(x[a:b].unsqueeze(0), pos[a:b].unsqueeze(0), tsc[a:b].unsqueeze(0), mem)
                pms.append(pm[0])
            _, ren = dec(x.unsqueeze(0), pos.unsqueeze(0), tsc.unsqueeze(0), mem, render=True)
            outs[mode] = torch.cat(pms + [ren[0]], 0).cpu()
            mems[mode] = mem
            assert mem[0][0].shape == (1, 48, 2 * cfg.dec_dim if mode == "kv" else cfg.dec_dim)
    finally:
        dec.change_memory_mode("kv")
    assert torch.equal(outs["norm_y"], outs["kv"])
    e_raw = rel_inf(outs["raw"], outs["kv"])
    # oracle memories in the two token modes
    xo, po = x.cpu(), pos.cpu()
    errs = {"raw_vs_kv": e_raw}
    for mode in ("norm_y", "raw"):
        memo = None
        for a, b in ((0, 2), (2, 3), (3, 4)):
            memo, _ = R.decoder_forward(sdd, cfg, xo[a:b].unsqueeze(0), po[a:b].unsqueeze(0), ts[a:b].unsqueeze(0), memo, False, mode)
        errs["mem_" + mode] = max(rel_inf(a.float().cpu(), b) for a, b in zip(mems[mode][0], memo[0]))
    record("memory_modes", precision=precision, **errs)
    u = 2.0 ** -8 if precision == "bf16" else 2.0 ** -11
    assert e_raw < TOL[precision] and errs["mem_norm_y"] < TOL[precision] + u and errs["mem_raw"] < TOL[precision] + u, errs


def test_baseline_mixed_resolution_forward_list():
    """BASELINE.json configs[4] geometry: MUSt3R_512 on a mixed-resolution batch 512x{384,336,160} through forward_list
    (one list entry per aspect ratio, shared memory), update + render, against the fp32 oracle."""
    from oracle import must3r_ref as R
    cfg = MUST3R_512
    enc, dec = build(cfg, "fp16w2")
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    sizes = [(384, 512), (336, 512), (160, 512)]
    imgs = [S.make_images(1, h, w, 10 + i) for i, (h, w) in enumerate(sizes)]
    xs, ps, ts, xo, po = [], [], [], [], []
    with torch.no_grad():
        for im, t in imgs:
            x, p = enc(im.cuda(), t.cuda())
            xs.append(x.unsqueeze(0)); ps.append(p.unsqueeze(0)); ts.append(t.cuda().unsqueeze(0))
            a, b = R.encoder_forward(sde, cfg, im, t, sdpa=True)
            xo.append(a.unsqueeze(0)); po.append(b.unsqueeze(0))
    mem, upd = dec(xs, ps, ts, None)
    _, ren = dec(xs, ps, ts, mem, render=True)
    tsc = [t.unsqueeze(0) for _, t in imgs]
    with torch.no_grad():
        memo, updo = R.decoder_forward(sdd, cfg, xo, po, tsc, None, False, "kv", sdpa=True)
        _, reno = R.decoder_forward(sdd, cfg, xo, po, tsc, memo, True, "kv", sdpa=True)
    errs = {}
    for i, (h, w) in enumerate(sizes):
        assert upd[i].shape == (1, 1, h, w, 7)
        errs[f"x_{h}"] = rel_inf(xs[i].cpu(), xo[i])
        errs[f"update_{h}"] = rel_inf(upd[i].cpu(), updo[i])
        errs[f"render_{h}"] = rel_inf(ren[i].cpu(), reno[i])
    record("baseline_mixed_resolution", **errs)
    assert mem[0][0].shape[1] == sum((h // 16) * (w // 16) for h, w in sizes)
    assert max(errs.values()) < TOL["fp16w2"], errs


def test_native_memory_surgery_keeps_append_in_place():
    """must3r_amd.engine.remove_from_mem (the in-place form of engine/inference.py:205-213) on device memory: the
    surviving tokens equal the reference's boolean-index result, the tensors remain views of the decoder's buffers, and
    the next update appends behind them without reallocating."""
    from must3r_amd.engine import remove_from_mem
    from oracle import must3r_ref as R
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    sdd = S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(4, 48, 64, 3)
    x, pos = enc(imgs.cuda(), ts.cuda())
    mem, _ = dec(x[:2].unsqueeze(0), pos[:2].unsqueeze(0), ts[:2].unsqueeze(0), None)
    mem, _ = dec(x[2:3].unsqueeze(0), pos[2:3].unsqueeze(0), ts[2:3].unsqueeze(0), mem)
    expect = [v[mem[1] != 1].view(1, -1, v.shape[-1]).clone() for v in mem[0]]
    base_ptr = mem[0][0].data_ptr()
    vals, labels = remove_from_mem(mem[0], mem[1], 1)
    assert all(torch.equal(a, b) for a, b in zip(vals, expect)) and vals[0].data_ptr() == base_ptr
    assert labels.tolist() == [[0] * 12 + [2] * 12]
    edited = (vals, labels, mem[2], mem[3], mem[4])
    mem2, _ = dec(x[3:4].unsqueeze(0), pos[3:4].unsqueeze(0), ts[3:4].unsqueeze(0), edited)
    assert mem2[0][0].data_ptr() == base_ptr and mem2[0][0].shape[1] == 36      # appended in place
    assert torch.equal(mem2[0][1][:, :24], expect[1])
    _, pm = dec(x.unsqueeze(0), pos.unsqueeze(0), ts.unsqueeze(0), mem2, render=True)
    mem_cpu = ([v.float().cpu() for v in mem2[0]], mem2[1].cpu(), mem2[2], mem2[3], mem2[4])
    _, ref = R.decoder_forward(sdd, cfg, x.cpu().unsqueeze(0), pos.cpu().unsqueeze(0), ts.unsqueeze(0), mem_cpu, True, "kv")
    assert rel_inf(pm.cpu(), ref) < TOL["fp16w2"]


def test_streaming_memory_schedule_vs_oracle():
    """BASELINE.json configs[3] semantics at test scale: online memory with a local window and keyframe-only retention
    (engine/inference.py:232-366), HIP path with in-place eviction vs the oracle driven through the SAME schedule with the
    reference's functional eviction.  12 frames, window 4, keyframe every 3rd."""
    from must3r_amd.engine import run_video
    from oracle import must3r_ref as R
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(12, 48, 64, 9)
    mem, pm0, kf = run_video(enc, dec, imgs.cuda(), ts, local_context_size=4)
    enc_o = lambda im, t: R.encoder_forward(sde, cfg, im, t)  # noqa: E731
    dec_o = lambda x, p, t, m=None, render=False: R.decoder_forward(sdd, cfg, x, p, t, m, render, "kv")  # noqa: E731
    with torch.no_grad():
        memo, pmo, kfo = run_video(enc_o, dec_o, imgs, ts, local_context_size=4)
    assert kf == kfo == [0, 1, 3, 6, 9]
    assert torch.equal(mem[1].cpu(), memo[1]) and mem[2:] == memo[2:]          # surviving labels: keyframes only
    assert mem[1].unique().tolist() == [0, 1, 3, 6, 9] and mem[0][0].shape[1] == 5 * 12
    e_pm = rel_inf(pm0.cpu(), pmo)
    e_mem = max(rel_inf(a.float().cpu(), b) for a, b in zip(mem[0], memo[0]))
    record("streaming_vs_oracle", pointmaps=e_pm, memory=e_mem)
    assert e_pm < TOL["fp16w2"] and e_mem < TOL["fp16w2"] + 2.0 ** -11, (e_pm, e_mem)
    # and the survivors can be rendered against
    _, ren = dec(*[t.unsqueeze(0) for t in enc(imgs[:2].cuda(), ts[:2])], ts[:2].unsqueeze(0), mem, render=True)
    assert torch.isfinite(ren).all()


def test_return_feats_matches_oracle():
    """decoder.py:344-347 / :258-262: [encoder tokens, residual stream after each block, norm_dec(last)], tensor and list
    inputs, update and render; the list dispatch of forward() drops the flag like the reference's (decoder.py:270)."""
    from oracle import must3r_ref as R
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(3, 48, 64, 4)
    x, pos = enc(imgs.cuda(), ts.cuda())
    xo, po = R.encoder_forward(sde, cfg, imgs, ts)
    tsc = ts.cuda()
    mem, pm, feats = dec(x[:2].unsqueeze(0), pos[:2].unsqueeze(0), tsc[:2].unsqueeze(0), None, return_feats=True)
    memo, pmo, featso = R.decoder_forward(sdd, cfg, xo[:2].unsqueeze(0), po[:2].unsqueeze(0), ts[:2].unsqueeze(0), None, False, "kv",
                                          return_feats=True)
    fo = featso
    assert len(feats) == cfg.dec_depth + 1 == len(fo)
    assert feats[0].shape == (1, 2, 12, cfg.enc_dim) and feats[-1].shape == (1, 2, 12, cfg.dec_dim)
    errs = [rel_inf(a.cpu().reshape(-1), torch.as_tensor(b).reshape(-1)) for a, b in zip(feats, fo)]
    record("return_feats", errs=errs)
    assert max(errs) < TOL["fp16w2"], errs
    # same pointmaps / memory as without the flag
    mem2, pm2 = dec(x[:2].unsqueeze(0), pos[:2].unsqueeze(0), tsc[:2].unsqueeze(0), None)
    assert torch.equal(pm, pm2) and all(torch.equal(a, b) for a, b in zip(mem[0], mem2[0]))
    # render + list form through forward_list; forward() with lists ignores the flag
    _, pml, fl = dec.forward_list([x[2:3].unsqueeze(0)], [pos[2:3].unsqueeze(0)], [tsc[2:3].unsqueeze(0)], mem, render=True, return_feats=True)
    _, pmr = dec(x[2:3].unsqueeze(0), pos[2:3].unsqueeze(0), tsc[2:3].unsqueeze(0), mem, render=True)
    assert torch.equal(pml[0], pmr) and len(fl) == 1 and len(fl[0]) == cfg.dec_depth + 1
    assert len(dec([x[2:3].unsqueeze(0)], [pos[2:3].unsqueeze(0)], [tsc[2:3].unsqueeze(0)], mem, render=True, return_feats=True)) == 2


def test_full_size_scene_properties():
    """BASELINE config 3 at full size (20 views, 384x512, MUSt3R_512) through size-independent properties: the batched
    encoder (M = 15360 rows: the 8-wave 256-row GEMM tiles) is bit-identical to per-view encoding (M = 768: 64x64 tiles),
    the scene's first update call equals the same call made alone, memory bookkeeping, render independence of a view
    from its batch, finiteness."""
    from must3r_amd.engine import run_scene
    cfg = MUST3R_512
    H, W, V = 384, 512, 20
    enc, dec = build(cfg, "fp16w2")
    imgs, ts = S.make_images(V, H, W, 0)
    imgs_c, ts_c = imgs.cuda(), ts.cuda()
    out = run_scene(enc, dec, imgs_c, ts_c)
    x, pos, mem = out["x"], out["pos"], out["mem"]
    for v in (0, 7, 19):
        xv, pv = enc(imgs_c[v:v + 1], ts_c[v:v + 1])
        assert torch.equal(xv[0], x[v]) and torch.equal(pv[0], pos[v]), f"encoder batch-variance at view {v}"
    assert out["update"].shape == out["render"].shape == (V, H, W, 7)
    assert all(torch.isfinite(out[k]).all() for k in ("update", "render", "pts3d", "pts3d_local", "conf"))
    assert (out["conf"] >= 1.0).all()
    assert mem[0][0].shape == (1, V * 768, 1536) and len(mem[0]) == cfg.dec_depth and mem[2:] == (V, V, V * 768)
    assert torch.equal(mem[1].cpu(), torch.arange(V).repeat_interleave(768).view(1, -1))
    m2, upd2 = dec(x[:2].unsqueeze(0), pos[:2].unsqueeze(0), ts_c[:2].unsqueeze(0), None)
    assert torch.equal(upd2[0], out["update"][:2])
    assert all(torch.equal(a[:, :2 * 768], b) for a, b in zip(mem[0], m2[0])), "the first 2 views' memory rows changed later"
    _, r1 = dec(x[11:12].unsqueeze(0), pos[11:12].unsqueeze(0), ts_c[11:12].unsqueeze(0), mem, render=True)
    e = rel_inf(r1[0, 0].cpu(), out["render"][11].cpu())
    record("full_size_render_independence", err=e)
    assert e < 0.5 * TOL["fp16w2"], e
    # two identical runs: bit-identical (no atomics / run-to-run nondeterminism anywhere on the path)
    out2 = run_scene(enc, dec, imgs_c, ts_c)
    assert torch.equal(out2["render"], out["render"]) and torch.equal(out2["update"], out["update"])


@pytest.mark.parametrize("fb", ["single_linear", None])
def test_feedback_types_vs_oracle(fb):
    """feedback_mechanism.py:11-22: the two feedback variants besides 'single_mlp' (update + update + render, kv memory)."""
    from oracle import must3r_ref as R
    from must3r_amd.model import Dust3rEncoder, MUSt3R
    cfg = TINY
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0, feedback_type=fb)
    enc, _ = build(cfg, "fp16w2")
    dec = MUSt3R(img_size=(cfg.img_size, cfg.img_size), enc_embed_dim=cfg.enc_dim, patch_size=cfg.patch_size, embed_dim=cfg.dec_dim,
                 output_dim=cfg.output_dim, depth=cfg.dec_depth, num_heads=cfg.dec_heads, feedback_type=fb, memory_mode="kv",
                 landscape_only=False, precision="fp16w2").cuda().eval()
    dec.load_state_dict(sdd, strict=True)
    imgs, ts = S.make_images(3, 48, 64, 1)
    x, pos = enc(imgs.cuda(), ts.cuda())
    tsc = ts.cuda()
    mem, pm = dec(x[:2].unsqueeze(0), pos[:2].unsqueeze(0), tsc[:2].unsqueeze(0), None)
    mem, pm2 = dec(x[2:].unsqueeze(0), pos[2:].unsqueeze(0), tsc[2:].unsqueeze(0), mem)
    _, ren = dec(x.unsqueeze(0), pos.unsqueeze(0), tsc.unsqueeze(0), mem, render=True)
    with torch.no_grad():
        xo, po = R.encoder_forward(sde, cfg, imgs, ts)
        memo, pmo = R.decoder_forward(sdd, cfg, xo[:2].unsqueeze(0), po[:2].unsqueeze(0), ts[:2].unsqueeze(0), None, False, "kv")
        memo, pmo2 = R.decoder_forward(sdd, cfg, xo[2:].unsqueeze(0), po[2:].unsqueeze(0), ts[2:].unsqueeze(0), memo, False, "kv")
        _, reno = R.decoder_forward(sdd, cfg, xo.unsqueeze(0), po.unsqueeze(0), ts.unsqueeze(0), memo, True, "kv")
    errs = [rel_inf(pm.cpu(), pmo), rel_inf(pm2.cpu(), pmo2), rel_inf(ren.cpu(), reno)]
    record("feedback_types", fb=str(fb), errs=errs)
    assert max(errs) < TOL["fp16w2"], errs


def test_render_of_more_views_than_one_view_table_holds():
    """The reference renders every view of an aspect ratio in ONE decoder call when no max_bs is given (engine/inference.py:489-522).
    The native call's view tables hold 1365 views; larger calls are cut into chunks of 1024 by the module (rendered views are
    independent).  2100 views of 48x64 against a 3-view memory: same pointmaps as rendering sampled views alone, and the current
    device of the caller is left alone by the native entry points."""
    enc, dec = build(TINY, "fp16w2")
    imgs, ts = S.make_images(3, 48, 64, 5)
    x, pos = enc(imgs.cuda(), ts.cuda())
    mem, _ = dec(x.unsqueeze(0), pos.unsqueeze(0), ts.unsqueeze(0), None)
    V = 2100
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = x[torch.randint(0, 3, (V,), device="cuda", generator=g)] + 0.05 * torch.randn((V,) + tuple(x.shape[1:]), device="cuda", generator=g)
    ps = pos[:1].expand(V, -1, -1).contiguous()
    tsv = ts[:1].expand(V, -1).contiguous()
    dev_before = torch.cuda.current_device()
    mem_r, ren = dec(xs.unsqueeze(0), ps.unsqueeze(0), tsv.unsqueeze(0), mem, render=True)
    assert torch.cuda.current_device() == dev_before
    assert ren.shape == (1, V, 48, 64, 7) and mem_r is mem and torch.isfinite(ren).all()
    for v in (0, 1023, 1024, 2047, 2048, V - 1):
        _, one = dec(xs[v:v + 1].unsqueeze(0), ps[v:v + 1].unsqueeze(0), tsv[v:v + 1].unsqueeze(0), mem, render=True)
        assert rel_inf(one[0, 0].cpu(), ren[0, v].cpu()) < 0.5 * TOL["fp16w2"], v


def test_sparse_low_part_batch_dependence_is_bounded():
    """fp16wa is NOT batch-invariant since r05: chip-filling split-weight launches multiply a 2:4-sparse low part of the weights,
    smaller launches the dense one.  Here the sparse path really runs (20 views of 384x512 = 15360 rows: every attention-side launch of the encoder takes it) and its
    distance from the dense low part is bounded by an assertion, two ways: the same 20-view batch with the switch off (SPARSE_LO = 0: dense two-pass kernels), and
    view 3 alone (a 768-row launch never takes the sparse path).  Both must stay far inside the mode's tolerance; everything else about the two runs is identical."""
    from must3r_amd import _lib
    cfg = MUST3R_512
    H, W, V = 384, 512, 20
    enc, _ = build(cfg, "fp16wa")
    imgs, ts = S.make_images(V, H, W, 0)
    imgs_c, ts_c = imgs.cuda(), ts.cuda()
    try:
        _lib.set_option("SPARSE_LO", 1)
        x_sp, _ = enc(imgs_c, ts_c)
        x_sp = x_sp.clone()
        x_one, _ = enc(imgs_c[3:4], ts_c[3:4])
        _lib.set_option("SPARSE_LO", 0)
        x_de, _ = enc(imgs_c, ts_c)
        x_one_de, _ = enc(imgs_c[3:4], ts_c[3:4])
    finally:
        _lib.set_option("SPARSE_LO", 1)
    assert torch.equal(x_one, x_one_de), "a one-view launch must not depend on the sparse switch"
    assert torch.equal(x_de[3], x_one_de[0]), "with dense low parts everywhere the encoder is batch-invariant again"
    e_ab = max(rel_inf(x_sp[v].cpu(), x_de[v].cpu()) for v in range(V))
    e_one = rel_inf(x_one[0].cpu(), x_sp[3].cpu())
    record("sparse_low_part_batch_dependence", sparse_vs_dense_20_views=e_ab, one_view_vs_view_of_20=e_one)
    assert e_ab > 0.0, "the sparse path did not run: the test is vacuous"
    # measured (r06, profiles/r06_test_metrics.jsonl): 5.7e-4 both ways on the encoder tokens -- the dropped (smaller) half of W_lo is worth about as much as the
    # mode's whole distance from the oracle (the two paths sit ~7e-4 from the fp32 oracle on different sides of it; each is asserted against the fixtures).
    # The bound keeps a batched and a one-at-a-time run of the same view inside ONE tolerance of each other.
    assert e_ab < 0.8 * TOL["fp16wa"] and e_one < 0.8 * TOL["fp16wa"], (e_ab, e_one)

