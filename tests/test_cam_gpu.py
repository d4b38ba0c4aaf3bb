"""GPU: postprocess(compute_cam=True) -- SURVEY.md section 8f rank 1 (engine/inference.py:29-47) -- through the C ABI
(must3r_hip_postprocess_cam) against the CPU oracle (oracle/cam_ref.py) and the reference-generated fixture.

Tolerances (floating point): focal within 2e-5 relative, c2w within 2e-5 of the scene scale.  The GPU path accumulates
its sums in fp64 (the reference in fp32) and uses v_rsq_f32 for the IRLS weight, so it is compared both with the fp32
restatement and with the fp64 evaluation of the same formulas."""
import pytest
import torch

from oracle import cam_ref, must3r_ref as R
from must3r_amd import synthetic as S
from util import load_golden
from test_ops_gpu import record

pytestmark = pytest.mark.gpu
F_RTOL, C_ATOL = 2e-5, 2e-5


def _check(pm_cpu, tag):
    from must3r_amd.engine import postprocess
    out = postprocess(pm_cpu.cuda(), compute_cam=True)
    act = R.postprocess(pm_cpu)
    for k in ("pts3d", "pts3d_local", "conf"):
        assert torch.allclose(out[k].cpu(), act[k], rtol=2e-6, atol=1e-6), k
    o32 = cam_ref.compute_cam(act["pts3d"], act["pts3d_local"], act["conf"])
    o64 = cam_ref.compute_cam(act["pts3d"], act["pts3d_local"], act["conf"], dtype=torch.float64)
    f, c = out["focal"].cpu(), out["c2w"].cpu()
    assert f.shape == o32["focal"].shape and c.shape == o32["c2w"].shape
    scale = max(1.0, float(o64["c2w"].abs().max()))
    ef32 = float(((f - o32["focal"]) / o32["focal"]).abs().max())
    ef64 = float(((f.double() - o64["focal"]) / o64["focal"]).abs().max())
    ec32 = float((c - o32["c2w"]).abs().max()) / scale
    ec64 = float((c.double() - o64["c2w"]).abs().max()) / scale
    record("cam_" + tag, focal_rel_vs_fp32=ef32, focal_rel_vs_fp64=ef64, c2w_vs_fp32=ec32, c2w_vs_fp64=ec64)
    assert ef32 < F_RTOL and ef64 < F_RTOL, (ef32, ef64)
    assert ec32 < C_ATOL and ec64 < C_ATOL, (ec32, ec64)
    R3 = c[..., :3, :3].double()
    assert torch.allclose(torch.det(R3), torch.ones(R3.shape[:-2], dtype=torch.float64), atol=1e-5)   # proper rotations
    assert torch.equal(c[..., 3, :], torch.tensor([0.0, 0.0, 0.0, 1.0]).expand_as(c[..., 3, :]))
    return out


def test_cam_reference_fixture():
    g = load_golden("cam_40x56")
    out = _check(torch.from_numpy(g["pm"]), "fixture")
    assert torch.allclose(out["focal"].cpu(), torch.from_numpy(g["focal"]), rtol=F_RTOL)
    assert torch.allclose(out["c2w"].cpu(), torch.from_numpy(g["c2w"]), atol=C_ATOL * 4)


@pytest.mark.parametrize("shape", [(1, 384, 512), (20, 384, 512), (30, 384, 512), (1, 3, 37, 53), (7, 224, 224),
                                   (2, 512, 512), (300, 16, 16)])
def test_cam_shapes(shape):
    """one view (G = 192 blocks), the bench batch (G = 12), more views than one cooperative launch holds, batch dims,
    pixel counts that are not multiples of the block, many tiny views."""
    pm = S.make_cam_pointmaps(*shape, focal=0.9 * max(shape[-2:]), noise=0.02, seed=sum(shape))
    _check(pm, "x".join(map(str, shape)))


def test_cam_degenerate_pixels_and_determinism():
    """z = 0 / inf ratios are dropped like nan_to_num does; zero-weight pixels; two runs are bit-identical."""
    from must3r_amd.engine import postprocess
    pm = S.make_cam_pointmaps(3, 64, 96, focal=70.0, noise=0.02, seed=5)
    pm[0, 3, 4, 3:6] = torch.tensor([0.3, -0.2, 0.0])
    pm[1, 10, 20, 3:6] = 0.0                        # 0/0 -> nan -> 0
    pm[2, :8, :, 6] = -200.0                        # conf - 1 == 0 exactly
    a = _check(pm, "degenerate")
    b = postprocess(pm.cuda(), compute_cam=True)
    assert torch.equal(a["focal"], b["focal"]) and torch.equal(a["c2w"], b["c2w"])


def test_cam_empty_and_errors():
    from must3r_amd.engine import postprocess
    out = postprocess(torch.zeros((0, 32, 32, 7), device="cuda"), compute_cam=True)
    assert out["focal"].shape == (0,) and out["c2w"].shape == (0, 4, 4)
    with pytest.raises(RuntimeError):
        postprocess(torch.zeros((1, 32, 32, 7)), compute_cam=True)      # CPU tensor: no fallback
