"""GPU tests added in round 4 (VERDICT r03 "Next round" item 5): the BENCHED configuration against the real-reference fixture, the
checkpoint -> load_model(device="cuda") -> forward route, the first RCCL execution of must3r_amd/parallel.py (a process group of
one rank on the box's one GPU) and ActivationType.LINEAR in the fused postprocess.
(File named to run after the older suites.)"""
import argparse
import os
import socket

import pytest
import torch

from must3r_amd import synthetic as S
from must3r_amd.config import TINY, SMALL, MUST3R_512
from util import TOL, TOL_DEFAULT_FIXTURE, load_golden, rel_inf, rel_inf_view
from test_model_gpu import build
from test_ops_gpu import record

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Sn", [8, 20, 28])   # 28 = bench.py's default --scenes (the configuration BENCH_rNN times)
def test_benched_configuration_scenes_in_flight_vs_reference_fixture(Sn):
    """What bench.py times -- S scenes of 20 views 384x512 IN FLIGHT in the default precision -- checked here, not only inside bench.py:
    scene 0 is the fixture's scene (outputs of the REAL reference, oracle/make_golden.py): every view of its update and render passes
    against the fixture, per view; every other scene against its own one-scene-at-a-time run (other tile shapes / split-KV factors /
    LN fold: within the mode's tolerance, not bit-equal)."""
    from must3r_amd.engine import run_scene, run_scenes
    import must3r_amd.model as M
    g = load_golden("must3r512_v20")
    H, W, V, ps, tks = (int(v) for v in g["meta"][:5])
    mb = [int(v) for v in g["meta"][5:]]
    import inspect
    prec = inspect.signature(M.MUSt3R.__init__).parameters["precision"].default   # the module default = bench.py's default
    enc, dec = build(MUST3R_512, prec)
    scenes = torch.stack([S.make_images(V, H, W, 0 if b == 0 else 500 + b)[0] for b in range(Sn)]).cuda()
    ts = S.make_images(V, H, W, 0)[1]
    out = run_scenes(enc, dec, scenes, ts, mem_batches=mb)
    torch.cuda.synchronize()
    upd, ren = out["update"][0].cpu(), out["render"][0].cpu()
    upd_v = [rel_inf_view(upd[v, ::ps, ::ps], g["update"][v], g["update_vmax"][v]) for v in range(V)]   # per-view FULL-resolution range (r05)
    ren_v = [rel_inf_view(ren[v, ::ps, ::ps], g["render"][v], g["render_vmax"][v]) for v in range(V)]
    upd_s = [rel_inf(upd[v, ::ps, ::ps], g["update"][v]) for v in range(V)]                              # the r04 figure (range of the sample), recorded only
    ren_s = [rel_inf(ren[v, ::ps, ::ps], g["render"][v]) for v in range(V)]
    others = []
    for b in range(1, min(Sn, 8)):   # (seven of the other scenes: each costs a one-scene-at-a-time run)
        one = run_scene(enc, dec, scenes[b], ts, mem_batches=mb)
        others.append(max(rel_inf(out["update"][b].cpu(), one["update"].cpu()), rel_inf(out["render"][b].cpu(), one["render"].cpu())))
    record("benched_configuration", precision=prec, scenes=Sn, update_per_view=[round(e, 6) for e in upd_v],
           render_per_view=[round(e, 6) for e in ren_v], others_vs_single=[round(e, 6) for e in others],
           update_worst_sampled_range=round(max(upd_s), 6), render_worst_sampled_range=round(max(ren_s), 6))
    assert torch.isfinite(out["render"]).all() and torch.isfinite(out["update"]).all()
    assert max(upd_v) < TOL_DEFAULT_FIXTURE and max(ren_v) < TOL_DEFAULT_FIXTURE, (prec, upd_v, ren_v)
    # r06: every pixel of the worst views against the REAL reference's full-resolution maps (oracle/make_golden.py full_views) -- the figure
    # bench.py reports against the port (config.parity.rel_inf_worst_view, 7.1-7.3e-4), asserted here against the reference itself at the benched S
    gf = load_golden("must3r512_v20_fullviews")
    full_u = [rel_inf(upd[int(v)], gf["update"][k]) for k, v in enumerate(gf["update_views"])]
    full_r = [rel_inf(ren[int(v)], gf["render"][k]) for k, v in enumerate(gf["render_views"])]
    record("benched_configuration_all_pixels", precision=prec, scenes=Sn, update_views=gf["update_views"].tolist(), render_views=gf["render_views"].tolist(),
           update=[round(e, 7) for e in full_u], render=[round(e, 7) for e in full_r])
    assert max(full_u + full_r) < TOL_DEFAULT_FIXTURE, (full_u, full_r)
    assert max(others) < TOL[prec], others


def _ckpt(tmp_path, cfg, img_size=None):
    ns = argparse.Namespace(
        encoder=f"Dust3rEncoder(img_size=({cfg.img_size},{cfg.img_size}),embed_dim={cfg.enc_dim},depth={cfg.enc_depth},num_heads={cfg.enc_heads})",
        decoder=(f"CausalMUSt3R(img_size=({cfg.img_size}, {cfg.img_size}), enc_embed_dim={cfg.enc_dim}, embed_dim={cfg.dec_dim}, "
                 f"depth={cfg.dec_depth}, num_heads={cfg.dec_heads}, feedback_type='single_mlp', memory_mode='kv', mem_dropout=0.1)"))
    p = tmp_path / "ckpt.pth"
    torch.save({"args": ns, "encoder": S.make_encoder_state_dict(cfg, 0), "decoder": S.make_decoder_state_dict(cfg, 0)}, p)
    return str(p)


def test_checkpoint_to_load_model_to_forward_on_the_gpu(tmp_path):
    """must3r/model/__init__.py:30-50: a checkpoint file (constructor strings + two state dicts) -> load_model(device='cuda') -> a scene,
    bit-equal to directly constructed modules holding the same weights; with ``img_size`` the RoPE base is rescaled
    (set_image_size_in_args, :66-108) and the result is the oracle's with F0 = trained / new."""
    import must3r_amd.model as M
    from must3r_amd.engine import run_scene
    from oracle import must3r_ref as R
    cfg = SMALL
    path = _ckpt(tmp_path, cfg)
    enc, dec = M.load_model(path, device="cuda", verbose=False)
    assert next(enc.parameters()).is_cuda and next(dec.parameters()).is_cuda and not enc.training and not dec.training
    assert dec.landscape_only is False and dec.memory_mode == "kv"
    imgs, ts = S.make_images(3, 224, 224, 4)
    got = run_scene(enc, dec, imgs.cuda(), ts)
    e2, d2 = build(cfg, dec.precision)
    want = run_scene(e2, d2, imgs.cuda(), ts)
    for k in ("update", "render", "pts3d", "conf"):
        assert torch.equal(got[k], want[k]), k
    assert all(torch.equal(a, b) for a, b in zip(got["mem"][0], want["mem"][0]))
    # memory_mode override (demo/gradio.py:58-59) and a second size: F0 = 224 / 448 in every RoPE of both halves
    enc3, dec3 = M.load_model(path, device="cuda", img_size=448, memory_mode="norm_y", verbose=False)
    assert dec3.memory_mode == "norm_y" and dec3.cfg.rope_f0 == 0.5 and enc3.cfg.rope_f0 == 0.5
    got3 = run_scene(enc3, dec3, imgs.cuda(), ts)
    import dataclasses
    cfg3 = dataclasses.replace(cfg, img_size=448, rope_f0=0.5)
    with torch.no_grad():
        upd_o, ren_o, _ = R.run_scene(S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0), cfg3, imgs, ts, memory_mode="norm_y")
    e_u, e_r = rel_inf(got3["update"].cpu(), upd_o), rel_inf(got3["render"].cpu(), ren_o)
    record("load_model_gpu", precision=dec3.precision, update=e_u, render=e_r)
    assert e_u < TOL[dec3.precision] and e_r < TOL[dec3.precision], (e_u, e_r)


def test_rccl_process_group_of_one_rank_runs_the_sharded_scene():
    """First RCCL execution of must3r_amd/parallel.py: init_process_group('nccl', world_size=1) on the box's GPU, then the view-sharded
    scene driver -- its all_gather_into_tensor of the encoded keyframe tokens really goes through RCCL (a group of one rank still issues
    the collective) -- must equal engine.run_scene on the same keyframes, bit for bit, with 16-bit tokens on the wire and without."""
    import torch.distributed as dist
    from must3r_amd.engine import run_scene
    from must3r_amd.parallel import all_gather_varlen, run_scene_sharded, run_video_sharded
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        t = torch.arange(12, dtype=torch.float32, device="cuda").view(3, 4)
        assert torch.equal(all_gather_varlen(t), t)                       # through ncclAllGather, not the world-1 shortcut of r03
        cfg = SMALL
        enc, dec = build(cfg, "fp16w2")
        V, H, W = 5, 224, 224
        imgs, ts = S.make_images(V, H, W, 11)
        kf = torch.tensor([True, True, False, True, False])
        out = run_scene_sharded(enc, dec, imgs.cuda(), ts, kf, keyframe_counts=[3], gather_outputs=True)
        torch.cuda.synchronize()
        # single-process equivalent: the memory from the keyframes, every view rendered against it
        x, pos = enc(imgs.cuda(), ts)
        sel = torch.nonzero(kf).flatten().cuda()
        one = run_scene(enc, dec, imgs.cuda()[sel], ts[kf], activate=False)
        _, ren = dec(x.unsqueeze(0), pos.unsqueeze(0), ts.unsqueeze(0), one["mem"], render=True)
        assert out["n_keyframes"] == 3 and torch.equal(out["render"], ren[0]) and torch.equal(out["render_all"], ren[0])
        assert all(torch.equal(a, b) for a, b in zip(out["mem"][0], one["mem"][0])) and torch.equal(out["mem"][1], one["mem"][1])
        # 16-bit tokens on the wire: the decoder rounds its operands to fp16 anyway -> the same bits
        out16 = run_scene_sharded(enc, dec, imgs.cuda(), ts, kf, comm_dtype=torch.float16, keyframe_counts=[3])
        assert torch.equal(out16["render"], out["render"])
        # streaming form
        ov = run_video_sharded(enc, dec, imgs.cuda(), ts, local_context_size=3, is_keyframe=lambda i: i % 2 == 0, frame_counts=[V])
        from must3r_amd.engine import run_video
        memv, pm0, kfs = run_video(enc, dec, imgs.cuda(), ts, local_context_size=3, is_keyframe=lambda i: i % 2 == 0)
        assert ov["keyframes"] == kfs and torch.equal(ov["pointmaps_0"], pm0) and torch.equal(ov["mem"][1], memv[1])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("compute_cam", [False, True])
def test_linear_activation_in_the_fused_postprocess(compute_cam):
    """ActivationType.LINEAR (must3r/model/blocks/head.py:13-21): the coordinates pass through, conf = 1 + exp(ch 6) as ever
    (engine/inference.py:26-27); with compute_cam the focal / registration run on the un-activated maps, like the reference."""
    from must3r_amd.engine import postprocess
    from must3r_amd.model import ActivationType
    from oracle import cam_ref
    pm = S.make_cam_pointmaps(3, 64, 96, focal=70.0, noise=0.02, seed=2) if compute_cam else torch.randn(2, 3, 40, 56, 7)
    out = postprocess(pm.cuda(), pointmaps_activation=ActivationType.LINEAR, compute_cam=compute_cam)
    assert torch.equal(out["pts3d"].cpu(), pm[..., 0:3]) and torch.equal(out["pts3d_local"].cpu(), pm[..., 3:6])
    assert torch.allclose(out["conf"].cpu(), 1.0 + pm[..., 6].exp(), rtol=2e-6, atol=1e-6)
    out_s = postprocess(pm.cuda(), pointmaps_activation="linear", compute_cam=compute_cam)          # the string form of head.py:14-15
    assert all(torch.equal(out[k], out_s[k]) for k in out)
    with pytest.raises(ValueError):
        postprocess(pm.cuda(), pointmaps_activation="tanh")
    if compute_cam:
        o64 = cam_ref.compute_cam(pm[..., 0:3], pm[..., 3:6], 1.0 + pm[..., 6].exp(), dtype=torch.float64)
        f, c = out["focal"].cpu().double(), out["c2w"].cpu().double()
        assert float(((f - o64["focal"]) / o64["focal"]).abs().max()) < 2e-5
        assert float((c - o64["c2w"]).abs().max()) / max(1.0, float(o64["c2w"].abs().max())) < 2e-5
    # NORM_EXP is untouched by the new argument
    a = postprocess(pm.cuda(), compute_cam=compute_cam)
    b = postprocess(pm.cuda(), pointmaps_activation=ActivationType.NORM_EXP, compute_cam=compute_cam)
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_mixed_resolution_scenes_in_flight_equal_their_single_scene_runs():
    """engine.run_scenes_mixed (BASELINE.json configs[4] with S scenes riding forward_list's batch dimension) against engine.run_scene_mixed
    of every scene alone and against the oracle's list path."""
    from must3r_amd.engine import run_scene_mixed, run_scenes_mixed
    from oracle import must3r_ref as R
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    shapes = [(2, 48, 64), (1, 32, 64), (2, 64, 64)]
    Sn = 3
    groups = [torch.stack([S.make_images(n, h, w, 70 + 10 * b + gi)[0] for b in range(Sn)]) for gi, (n, h, w) in enumerate(shapes)]
    out = run_scenes_mixed(enc, dec, [g.cuda() for g in groups])
    torch.cuda.synchronize()
    assert [tuple(r.shape) for r in out["render"]] == [(Sn, n, h, w, 7) for n, h, w in shapes]
    worst = 0.0
    for b in range(Sn):
        one = run_scene_mixed(enc, dec, [g[b].cuda() for g in groups])
        for gi in range(len(shapes)):
            worst = max(worst, rel_inf(out["render"][gi][b].cpu(), one["render"][gi].cpu()))
            assert torch.allclose(out["pts3d"][gi][b], one["pts3d"][gi], rtol=1e-2, atol=1e-3)
        for v, u in enumerate(one["update"]):
            worst = max(worst, rel_inf(out["update"][v][b].cpu(), u.cpu()))
    # oracle of scene 1: encode per group, update [2,1,1,1] over the view order (group order, then order in the group), render all groups
    b = 1
    xs = [R.encoder_forward(sde, cfg, g[b], torch.tensor([[g.shape[-2], g.shape[-1]]] * g.shape[1])) for g in groups]
    tss = [torch.tensor([[g.shape[-2], g.shape[-1]]] * g.shape[1]) for g in groups]
    U = lambda t: t.unsqueeze(0)  # noqa: E731
    mem, _ = R.decoder_forward(sdd, cfg, U(xs[0][0]), U(xs[0][1]), U(tss[0]), None, False, "kv")
    mem, _ = R.decoder_forward(sdd, cfg, U(xs[1][0]), U(xs[1][1]), U(tss[1]), mem, False, "kv")
    for j in range(2):
        mem, _ = R.decoder_forward(sdd, cfg, U(xs[2][0][j:j + 1]), U(xs[2][1][j:j + 1]), U(tss[2][j:j + 1]), mem, False, "kv")
    err = 0.0
    for gi in range(len(shapes)):
        _, ro = R.decoder_forward(sdd, cfg, U(xs[gi][0]), U(xs[gi][1]), U(tss[gi]), mem, True, "kv")
        err = max(err, rel_inf(out["render"][gi][b].cpu(), torch.as_tensor(ro)[0]))
    record("mixed_scenes_in_flight", vs_single=worst, vs_oracle=err)
    assert worst < TOL["fp16w2"] and err < TOL["fp16w2"], (worst, err)
