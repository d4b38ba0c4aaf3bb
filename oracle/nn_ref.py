"""TEST INFRASTRUCTURE -- CPU restatement of the SLAM keyframe test's overlap score.

Only ``tests/`` may import this file.  It follows, line by line:
  * must3r/slam/tools.py:9-31   ``get_quadrant_id`` (view-direction quadrants, float32 numpy like the reference's input)
  * must3r/slam/nns.py:40-57    ``KDTree_scipy`` (scipy.spatial.KDTree, k = 1, Euclidean; +inf on an empty tree)
  * must3r/slam/nns.py:60-92    ``QuandrantSearcher`` (one search structure per quadrant of the ray p - cam_center)
  * must3r/slam/model.py:62-91  ``get_overlap_score`` (conf mask, optional x/y subsampling, 'nn' / 'nn-norm', percentile)
PINNED: ``tests/test_oracle_vs_reference.py::test_overlap_score_equals_reference`` runs the reference's own modules
(they import verbatim: numpy + scipy only) against this file; ``tests/golden/nn_overlap.npz`` holds reference outputs.
scipy is the reference's own dependency for this component (nns.py:2) and is present in the image.
"""
import numpy as np
from scipy.spatial import KDTree


def get_quadrant_id(rays, quadrant_divider=4, eps=1e-5):
    rays = rays / np.linalg.norm(rays, axis=-1, keepdims=True).clip(eps)
    thetas = np.arccos(rays[:, -1]) / np.pi
    phis = np.arctan2(rays[:, 1], rays[:, 0]) / np.pi
    thetas = thetas.clip(eps, 1 - eps)
    phis = phis.clip(-1 + eps, 1 - eps)
    theta_idx = np.floor(thetas * quadrant_divider).astype(int)
    phis_idx = np.floor(phis * quadrant_divider).astype(int) + quadrant_divider
    return (theta_idx + phis_idx * quadrant_divider).astype(int)


class KDTreeSearcher:
    def __init__(self):
        self.all_points = []
        self.kdtree = None

    def add_pts(self, pts, **kw):
        pts = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
        self.all_points = pts if len(self.all_points) == 0 else np.concatenate([self.all_points, pts])
        self.kdtree = KDTree(self.all_points)

    def query(self, pts, **kw):
        pts = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
        if self.kdtree is None:
            return np.full(pts.shape[0], np.inf)
        return self.kdtree.query(pts, k=1, workers=4)[0]


class QuadrantSearcher:
    def __init__(self, quadrant_divider):
        self.quadrant_divider = quadrant_divider
        self.search_structs = [KDTreeSearcher() for _ in range(2 * quadrant_divider ** 2)]

    def add_pts(self, pts, cam_center, **kw):
        pts = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
        qid = get_quadrant_id(pts - np.asarray(cam_center, dtype=np.float32)[None], self.quadrant_divider)
        for quad in np.unique(qid):
            self.search_structs[quad].add_pts(pts[qid == quad])

    def query(self, pts, cam_center, **kw):
        pts = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
        qid = get_quadrant_id(pts - np.asarray(cam_center, dtype=np.float32)[None], self.quadrant_divider)
        dists = np.zeros(pts.shape[0])
        for quad in np.unique(qid):
            idx = qid == quad
            dists[idx] = self.search_structs[quad].query(pts[idx])
        return dists


def get_searcher(method):
    """nns.py:9-19."""
    if "quadrant_x" in method:
        return QuadrantSearcher(int(method.split("quadrant_x")[-1].split("-")[0]))
    if "kdtree-scipy" in method:
        return KDTreeSearcher()
    if method == "none":
        return None
    raise ValueError(f"Unknown searcher method {method}")


def get_overlap_score(res, overlap_tree, cam_center, mode="nn", kf_x_subsamp=None, min_conf_keyframe=1.5, percentile=70, eps=1e-9):
    """slam/model.py:62-91 on numpy inputs: res['pts3d'] [1,1,H,W,3], res['conf'] [1,1,H,W], res['pts3d_local'] [1,1,H,W,3]."""
    if mode == "meanconf":
        return res["conf"].mean()
    if mode == "medianconf":
        return np.median(res["conf"])
    if "nn" not in mode:
        raise ValueError(f"Unknown overlap score method {mode}")
    pts3d = res["pts3d"][0, 0, ::kf_x_subsamp, ::kf_x_subsamp] if kf_x_subsamp else res["pts3d"]
    msk = res["conf"][0, 0, ::kf_x_subsamp, ::kf_x_subsamp] if kf_x_subsamp else res["conf"]
    msk = msk > min_conf_keyframe
    outscore = 0.0
    if msk.sum() > 0:
        dists = np.array(overlap_tree.query(pts3d[msk], cam_center=cam_center), dtype=np.float64)
        if "norm" in mode:
            depths = res["pts3d_local"][0, 0, ::kf_x_subsamp, ::kf_x_subsamp, -1]
            dists /= depths[msk] + eps
        dists[np.isposinf(dists)] = np.finfo(dists.dtype).max
        outscore = np.percentile(dists, percentile)
    return outscore
