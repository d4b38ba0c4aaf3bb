"""SLAM keyframe test on the GPU -- drop-in for must3r/slam/nns.py and ``get_overlap_score`` (must3r/slam/model.py:62-91).

Same class / function names and call signatures as the reference (``get_searcher``, ``Base_NN.add_pts`` / ``.query``,
``QuandrantSearcher`` [sic], ``get_overlap_score``), torch CUDA tensors in, numpy distances / a float score out like the
reference.  Instead of rebuilding a scipy KD-tree over all keyframe points after every keyframe (nns.py:47-50) and
querying it on 4 host threads (nns.py:56), the points stay in HBM and a query is an exact brute-force scan
(``must3r_hip_nn_query``); quadrant ids (slam/tools.py:9-31) come from ``must3r_hip_quadrant_ids``.
``forward_must3r`` is the SLAM agent's forward wrapper (slam/model.py:22-59).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _check_pts(pts):
    if not isinstance(pts, torch.Tensor) or not pts.is_cuda:
        raise RuntimeError("must3r_amd.slam_nn: points must be CUDA tensors; the HIP path has no CPU fallback")
    return pts.reshape(-1, 3).float().contiguous()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def nn_distances(db, q):
    """Exact 1-NN Euclidean distances of q [n,3] to db [m,3] (cuda fp32) -> cuda fp32 [n]; +inf if db is empty."""
    q = _check_pts(q)
    out = torch.empty((q.shape[0],), dtype=torch.float32, device=q.device)
    n_db = 0 if db is None else int(db.shape[0])
    with torch.cuda.device(q.device):
        _lib.check(_lib.load().must3r_hip_nn_query(db.data_ptr() if n_db else None, n_db, q.data_ptr(), q.shape[0], out.data_ptr(), _stream(q)))
    return out


def quadrant_ids(pts, cam_center, quadrant_divider):
    """slam/tools.py:9-31 on rays pts - cam_center -> cuda int32 [n]."""
    pts = _check_pts(pts)
    out = torch.empty((pts.shape[0],), dtype=torch.int32, device=pts.device)
    cc = (C.c_float * 3)(*[float(v) for v in torch.as_tensor(cam_center).reshape(3).tolist()])
    with torch.cuda.device(pts.device):
        _lib.check(_lib.load().must3r_hip_quadrant_ids(pts.data_ptr(), pts.shape[0], cc, int(quadrant_divider), out.data_ptr(), _stream(pts)))
    return out


class Base_NN:
    """nns.py:22-38."""

    def __init__(self, subsamp=None):
        self.subsamp = subsamp

    def add_pts(self, pts, **kw):
        raise NotImplementedError("Overload this function for your needs")

    def query(self, pts, **kw):
        raise NotImplementedError("Overload this function for your needs")


class BruteForce_hip(Base_NN):
    """Stands in for KDTree_scipy (nns.py:40-57): same results (exact 1-NN), the database lives in HBM."""

    def __init__(self):
        super().__init__()
        self.all_points = None      # cuda fp32 [capacity, 3]; rows [0, n) valid
        self.n = 0

    def add_pts(self, pts, **kw):
        pts = _check_pts(pts)
        need = self.n + pts.shape[0]
        if self.all_points is None or need > self.all_points.shape[0]:
            grown = torch.empty((max(need, 2 * (0 if self.all_points is None else self.all_points.shape[0])), 3), dtype=torch.float32,
                                device=pts.device)
            if self.n:
                grown[:self.n] = self.all_points[:self.n]
            self.all_points = grown
        self.all_points[self.n:need] = pts
        self.n = need

    def query_device(self, pts, **kw):
        return nn_distances(None if self.n == 0 else self.all_points[:self.n], pts)

    def query(self, pts, **kw):
        return self.query_device(pts).double().cpu().numpy()


KDTree_scipy = BruteForce_hip   # the reference's name, for callers that construct it directly


class QuandrantSearcher(Base_NN):
    """nns.py:60-92 (spelling as in the reference): one search structure per view-direction quadrant."""

    def __init__(self, method):
        super().__init__()
        self.quadrant_divider = int(method.split("quadrant_x")[-1].split("-")[0])
        self.search_structs = [BruteForce_hip() for _ in range(2 * self.quadrant_divider ** 2)]

    def add_pts(self, pts, cam_center, **kw):
        pts = _check_pts(pts)
        qid = quadrant_ids(pts, cam_center, self.quadrant_divider)
        for quad in torch.unique(qid).tolist():
            self.search_structs[quad].add_pts(pts[qid == quad])

    def query_device(self, pts, cam_center, **kw):
        pts = _check_pts(pts)
        qid = quadrant_ids(pts, cam_center, self.quadrant_divider)
        dists = torch.zeros((pts.shape[0],), dtype=torch.float32, device=pts.device)
        for quad in torch.unique(qid).tolist():
            idx = qid == quad
            dists[idx] = self.search_structs[quad].query_device(pts[idx])
        return dists

    def query(self, pts, cam_center, **kw):
        return self.query_device(pts, cam_center).double().cpu().numpy()


def get_searcher(method, isquadrant=False):
    """nns.py:9-19; the reference's method strings ('kdtree-scipy', 'kdtree-scipy-quadrant_x2', 'none') are accepted."""
    if "quadrant_x" in method and not isquadrant:
        return QuandrantSearcher(method)
    if "kdtree-scipy" in method or "bruteforce-hip" in method:
        return BruteForce_hip()
    if method == "none":
        return None
    raise ValueError(f"Unknown searcher method {method}")


def get_overlap_score(res, overlap_tree, cam_center, mode="nn", kf_x_subsamp=None, min_conf_keyframe=1.5, percentile=70, eps=1e-9):
    """slam/model.py:62-91.  ``res`` holds CUDA tensors (the dict ``postprocess`` returns, with the leading [1,1] dims)."""
    outscore = 0.0
    if mode == "meanconf":
        return res["conf"].mean()
    if mode == "medianconf":
        return res["conf"].median()
    if "nn" not in mode:
        raise ValueError(f"Unknown overlap score method {mode}")
    pts3d = res["pts3d"][0, 0, ::kf_x_subsamp, ::kf_x_subsamp] if kf_x_subsamp else res["pts3d"]
    msk = res["conf"][0, 0, ::kf_x_subsamp, ::kf_x_subsamp] if kf_x_subsamp else res["conf"]
    msk = msk > min_conf_keyframe
    if msk.sum() > 0:
        dists = np.array(overlap_tree.query(pts3d[msk], cam_center=cam_center), dtype=np.float64)
        if "norm" in mode:
            depths = res["pts3d_local"][0, 0, ::kf_x_subsamp, ::kf_x_subsamp, -1]
            dists /= depths[msk].cpu().numpy() + eps
        dists[np.isposinf(dists)] = np.finfo(dists.dtype).max   # unseen quadrant (slam/model.py:87-88)
        outscore = np.percentile(dists, percentile)
    return outscore


@torch.no_grad()
def forward_must3r(model, input_views, memory, render=False, device='cuda:0', postprocess=None):
    """slam/model.py:22-59: encode every view (dict with ``img`` [1,3,H,W] and ``true_shape`` [1,2], dust3r's view layout) on its own,
    ONE decoder call over the list (one group per view: views may differ in aspect ratio), activation per view.
    Returns ``(list of {pts3d, pts3d_local, conf} with a leading [1, 1] batch, new_memory)``.

    The (H, W) pairs stay on the host (both modules read them as host integers: no device round trip per view) and the
    allocator cache is left alone (the reference empties it after every frame, :50).  ``postprocess`` defaults to the
    fused native activation (``must3r_amd.engine.postprocess``)."""
    if postprocess is None:
        from .engine import postprocess
    from .model import get_pointmaps_activation
    encoder, decoder = model
    xs, poss, shapes = [], [], []
    for view in input_views:
        shape = view['true_shape']
        shape = (shape.cpu() if torch.is_tensor(shape) else torch.as_tensor(shape))[None]   # :31, kept on the host
        x, pos = encoder(view['img'].to(device), shape.view(-1, 2))
        xs.append(x[None])
        poss.append(pos[None])
        shapes.append(shape)
    new_memory, preds = decoder(xs, poss, shapes, memory, render=render)
    activation = get_pointmaps_activation(decoder, verbose=False)
    return [postprocess(pred, pointmaps_activation=activation) for pred in preds], new_memory
