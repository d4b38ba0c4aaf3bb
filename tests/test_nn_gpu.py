"""GPU: the SLAM keyframe test through the C ABI (must3r_hip_nn_query,
must3r_hip_quadrant_ids) and the drop-in classes of must3r_amd.slam_nn, against the scipy oracle (oracle/nn_ref.py) and
the reference-generated fixture.  Nearest-neighbour distances are exact up to fp32 rounding of (q - p)^2 (the KD-tree
works in float64 on the same float32 points): relative tolerance 2e-6; quadrant ids may differ from numpy's only for rays
that sit on a quadrant boundary (acosf / atan2f differ from numpy's by ulps)."""
import numpy as np
import pytest
import torch

from oracle import nn_ref
from must3r_amd import synthetic as S
from util import load_golden
from test_ops_gpu import record

pytestmark = pytest.mark.gpu
RTOL = 2e-6


def _close(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    return float(np.max(np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1e-30))) if fin.any() else 0.0


@pytest.mark.parametrize("shape", [(1, 1), (1000, 1), (1, 1000), (5000, 777), (2049, 4097), (300000, 12288)])
def test_nn_query_exact(shape):
    from must3r_amd.slam_nn import nn_distances
    nd, nq = shape
    g = torch.Generator().manual_seed(nd + nq)
    db = torch.randn((nd, 3), generator=g) * 2.0
    q = torch.randn((nq, 3), generator=g) * 2.0
    if nq > 10:
        q[3] = db[min(5, nd - 1)]                       # exact hit -> distance 0
    d = nn_distances(db.cuda(), q.cuda()).cpu().numpy()
    from scipy.spatial import KDTree
    ref = KDTree(db.numpy()).query(q.numpy(), k=1, workers=4)[0]
    e = _close(d, np.where(ref == 0, 0.0, ref)) if not (ref == 0).any() else _close(d[ref > 0], ref[ref > 0])
    record("nn_query", shape=shape, rel_err=e)
    assert e < RTOL, e
    assert (d[ref == 0] == 0).all()


def test_nn_query_empty_database_and_errors():
    from must3r_amd.slam_nn import nn_distances, BruteForce_hip
    q = torch.randn((17, 3)).cuda()
    assert torch.isinf(nn_distances(None, q)).all()                     # nns.py:53-54
    assert np.isposinf(BruteForce_hip().query(q)).all()
    assert nn_distances(torch.randn((5, 3)).cuda(), torch.zeros((0, 3)).cuda()).shape == (0,)
    with pytest.raises(RuntimeError):
        nn_distances(torch.randn((5, 3)), torch.randn((5, 3)))          # CPU tensors: no fallback


@pytest.mark.parametrize("div", [1, 2, 4])
def test_quadrant_ids_match_numpy(div):
    from must3r_amd.slam_nn import quadrant_ids
    g = torch.Generator().manual_seed(div)
    pts = torch.randn((20000, 3), generator=g) * 3.0
    pts[:6] = torch.tensor([[0, 0, 1.0], [0, 0, -1.0], [1.0, 0, 0], [-1.0, 0, 0], [0, 1.0, 0], [0.0, 0.0, 0.0]])  # poles, axes, origin
    cam = torch.tensor([0.1, -0.2, 0.3])
    ids = quadrant_ids(pts.cuda(), cam, div).cpu().numpy()
    ref = nn_ref.get_quadrant_id((pts - cam[None]).numpy().astype(np.float32), div)
    mism = int((ids != ref).sum())
    record("quadrant_ids", div=div, mismatches=mism)
    assert ids.min() >= 0 and ids.max() < 2 * div * div
    assert mism <= 4, mism                                               # boundary rays only


def test_overlap_score_reference_fixture():
    """The whole keyframe test on the fixture sequence: scores and distances of the REAL reference (tests/golden)."""
    from must3r_amd.slam_nn import get_searcher, get_overlap_score
    gold = load_golden("nn_overlap")
    frames = S.make_overlap_frames(7, n_kf=4, H=48, W=64)
    for method in ("kdtree-scipy", "kdtree-scipy-quadrant_x2"):
        tree = get_searcher(method)
        worst_s = worst_d = 0.0
        for i, f in enumerate(frames):
            res = {k: torch.from_numpy(f[k]).cuda() for k in ("pts3d", "pts3d_local", "conf")}
            cam = torch.from_numpy(f["cam"])
            for j, m in enumerate(("nn", "nn-norm")):
                sc = float(get_overlap_score(res, tree, cam, mode=m, kf_x_subsamp=2, percentile=70))
                ref = float(gold[method + "/scores"][i][j])
                worst_s = max(worst_s, abs(sc - ref) / max(abs(ref), 1e-30) if np.isfinite(ref) and ref < 1e300 else float(sc != ref))
            d = tree.query(res["pts3d"][0, 0, ::2, ::2].reshape(-1, 3), cam_center=cam)
            worst_d = max(worst_d, _close(d, gold[method + "/dists"][i]))
            tree.add_pts(res["pts3d"][0, 0][res["conf"][0, 0] > 1.5], cam_center=cam)
        record("overlap_fixture", method=method, score_rel=worst_s, dist_rel=worst_d)
        assert worst_d < RTOL and worst_s < 1e-5, (method, worst_s, worst_d)


def test_overlap_score_full_size_vs_oracle():
    """384x512 pointmaps, subsampling 2 (49 k queries per frame), 6 keyframes, quadrant searcher: GPU vs scipy oracle."""
    from must3r_amd.slam_nn import get_searcher, get_overlap_score
    frames = S.make_overlap_frames(3, n_kf=6, H=384, W=512)
    tg, to = get_searcher("kdtree-scipy-quadrant_x2"), nn_ref.get_searcher("kdtree-scipy-quadrant_x2")
    worst = 0.0
    for f in frames:
        res = {k: torch.from_numpy(f[k]).cuda() for k in ("pts3d", "pts3d_local", "conf")}
        res_n = {k: f[k] for k in ("pts3d", "pts3d_local", "conf")}
        sg = float(get_overlap_score(res, tg, torch.from_numpy(f["cam"]), mode="nn-norm", kf_x_subsamp=2))
        so = float(nn_ref.get_overlap_score(res_n, to, f["cam"], mode="nn-norm", kf_x_subsamp=2))
        if np.isfinite(so) and so < 1e300:
            worst = max(worst, abs(sg - so) / max(abs(so), 1e-30))
        else:
            assert sg == so
        sel = f["pts3d"][0, 0, ::2, ::2][f["conf"][0, 0, ::2, ::2] > 1.5]
        tg.add_pts(torch.from_numpy(sel).cuda(), cam_center=torch.from_numpy(f["cam"]))
        to.add_pts(sel, cam_center=f["cam"])
    record("overlap_full_size", score_rel=worst)
    assert worst < 1e-4, worst      # a handful of boundary rays may land in the neighbouring quadrant
