"""TEST INFRASTRUCTURE -- CPU restatement of ``postprocess(compute_cam=True)``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline/parity leg may import this file.

What it follows:
  * the glue -- must3r/engine/inference.py:29-47: principal point ``(W/2, H/2)``, focal from ``pts3d_local`` with
    ``focal_mode='weiszfeld'``, rigid registration ``pts3d_local -> pts3d`` weighted by ``conf - 1``, 4x4 ``c2w``.
    PINNED: ``tests/test_oracle_vs_reference.py`` runs the reference's own ``postprocess`` (verbatim import) with the
    two leaves below registered as its ``dust3r.post_process`` / ``roma`` modules and compares.
  * the two leaves are THIRD-PARTY and un-vendored (not in /root/reference, no network) -- **parity unpinned**:
      - ``dust3r.post_process.estimate_focal_knowing_depth`` (dust3r, branch ``dust3r_setup`` per the reference's
        setup.py:37): published algorithm = closed-form L2 focal, then 10 IRLS (Weiszfeld) steps with weights
        1/max(dis, 1e-8), pixel grid ``xy_grid(W, H)`` (integer pixel coordinates, no half-pixel offset) minus ``pp``,
        result clipped to ``[min_focal, max_focal] * focal_base`` (defaults 0 and inf: no-op).
      - ``roma.rigid_points_registration(x, y, weights, compute_scaling=False)`` (roma, version unpinned in the
        reference's requirements): weighted Kabsch -- normalised weights, weighted centroids,
        ``M = sum_i w_i (y_i - ybar)(x_i - xbar)^T``, ``R = special_procrustes(M) = U diag(1,1,det(U V^T)) V^T``,
        ``t = ybar - R xbar``.
    They are restated from the published algorithms; the reference holds no test or golden vector for either.
"""
import math

import torch


def xy_grid(W, H, device=None):
    """dust3r.utils.geometry.xy_grid: [H, W, 2] with (x, y) integer pixel coordinates."""
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    return torch.stack((xs, ys), dim=-1)


def estimate_focal_knowing_depth(pts3d, pp, focal_mode="weiszfeld", min_focal=0.0, max_focal=float("inf"), n_iter=10):
    """pts3d [B,H,W,3] camera-frame points, pp [2] -> focal [B] (dust3r post_process, Weiszfeld branch)."""
    assert focal_mode == "weiszfeld", focal_mode
    B, H, W, three = pts3d.shape
    assert three == 3
    pixels = xy_grid(W, H, device=pts3d.device).view(1, -1, 2) - pp.view(-1, 1, 2)
    pts = pts3d.flatten(1, 2)
    xy_over_z = (pts[..., :2] / pts[..., 2:3]).nan_to_num(posinf=0, neginf=0)
    dot_xy_px = (xy_over_z * pixels).sum(dim=-1)
    dot_xy_xy = xy_over_z.square().sum(dim=-1)
    focal = dot_xy_px.mean(dim=1) / dot_xy_xy.mean(dim=1)
    for _ in range(n_iter):
        dis = (pixels - focal.view(-1, 1, 1) * xy_over_z).norm(dim=-1)
        w = dis.clip(min=1e-8).reciprocal()
        focal = (w * dot_xy_px).mean(dim=1) / (w * dot_xy_xy).mean(dim=1)
    focal_base = max(H, W) / (2 * math.tan(math.radians(60) / 2))
    return focal.clip(min=min_focal * focal_base, max=max_focal * focal_base)


def special_procrustes(M):
    """roma.special_procrustes: rotation R maximising trace(R^T M) (Kabsch/Umeyama sign fix)."""
    U, _, Vh = torch.linalg.svd(M)
    d = torch.det(U @ Vh)
    D = torch.ones(M.shape[:-1], dtype=M.dtype, device=M.device)
    D[..., -1] = d
    return (U * D.unsqueeze(-2)) @ Vh


def rigid_points_registration(x, y, weights=None, compute_scaling=False):
    """x, y [...,n,3]; weights [...,n] -> (R [...,3,3], t [...,3]) minimising sum_i w_i |R x_i + t - y_i|^2."""
    assert not compute_scaling
    if weights is None:
        xmean = x.mean(dim=-2, keepdim=True)
        ymean = y.mean(dim=-2, keepdim=True)
        xhat, yhat = x - xmean, y - ymean
        M = torch.einsum("...ki,...kj->...ij", yhat, xhat)
    else:
        w = weights / weights.sum(dim=-1, keepdim=True)
        xmean = (w[..., None] * x).sum(dim=-2, keepdim=True)
        ymean = (w[..., None] * y).sum(dim=-2, keepdim=True)
        xhat, yhat = x - xmean, y - ymean
        M = torch.einsum("...ki,...kj->...ij", w[..., None] * yhat, xhat)
    R = special_procrustes(M)
    t = ymean.squeeze(-2) - (R @ xmean.squeeze(-2)[..., None]).squeeze(-1)
    return R, t


def compute_cam(pts3d, pts3d_local, conf, dtype=torch.float32):
    """engine/inference.py:29-47 on activated maps [...,H,W,3] / [...,H,W] -> dict(focal [...], c2w [...,4,4]).

    ``dtype=torch.float64`` evaluates the same formulas in double precision (used as the accuracy yardstick: the GPU
    path accumulates in fp64, the reference in fp32)."""
    batch_dims = pts3d.shape[:-3]
    H, W = conf.shape[-2:]
    p3 = pts3d.to(dtype)
    pl = pts3d_local.to(dtype)
    cf = conf.to(dtype)
    pp = torch.tensor((W / 2, H / 2), dtype=dtype)
    if dtype == torch.float32:
        focal = estimate_focal_knowing_depth(pl.reshape(-1, H, W, 3), pp)
    else:
        # same algorithm, double precision grid
        B = math.prod(batch_dims) if batch_dims else 1
        pix = xy_grid(W, H).to(dtype).view(1, -1, 2) - pp.view(-1, 1, 2)
        pts = pl.reshape(B, -1, 3)
        q = (pts[..., :2] / pts[..., 2:3]).nan_to_num(posinf=0, neginf=0)
        a = (q * pix).sum(-1)
        b = q.square().sum(-1)
        focal = a.mean(1) / b.mean(1)
        for _ in range(10):
            dis = (pix - focal.view(-1, 1, 1) * q).norm(dim=-1)
            w = dis.clip(min=1e-8).reciprocal()
            focal = (w * a).mean(1) / (w * b).mean(1)
    R, T = rigid_points_registration(pl.reshape(*batch_dims, -1, 3), p3.reshape(*batch_dims, -1, 3),
                                     weights=cf.reshape(*batch_dims, -1) - 1.0)
    c2w = torch.eye(4, dtype=dtype).view(*([1] * len(batch_dims)), 4, 4).repeat(*batch_dims, 1, 1)
    c2w[..., :3, :3] = R
    c2w[..., :3, 3] = T.view(*batch_dims, 3)
    return {"focal": focal.reshape(*batch_dims), "c2w": c2w}
