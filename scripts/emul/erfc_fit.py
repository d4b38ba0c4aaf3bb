"""Fit of erfc(z) ~ 2^(-z P(z)) (the GELU epilogue of csrc/common.hpp, r04): least squares on -log2(erfc(z)) / z, reweighted towards the minimax of the absolute\nerfc error; prints the coefficients and the GELU error of an fp32 evaluation.  Experiment infrastructure, never imported by the product."""
import numpy as np
from scipy.special import erf as serf, erfc as serfc
from scipy.optimize import least_squares
def run(n, zmax):
    z = (np.cos(np.linspace(0, np.pi, 3001)) * 0.5 + 0.5) * zmax
    z = z[z > 1e-6]
    target = -np.log2(serfc(z)) / z            # L(z)
    A = np.stack([z**k for k in range(n)], axis=1)
    c0, *_ = np.linalg.lstsq(A, target, rcond=None)
    def resid(c):                              # abs error of erf (= of erfc)
        return np.exp2(-z * (A @ c)) - serfc(z)
    # minimax-ish: IRLS on the abs erfc error
    w = np.ones_like(z); c = c0.copy(); best = (1e9, c)
    for it in range(40):
        r = least_squares(lambda cc: resid(cc) * w, c, xtol=1e-15, ftol=1e-15, gtol=1e-15)
        c = r.x
        e = np.abs(resid(c)); m = e.max()
        if m < best[0]: best = (m, c.copy())
        w = w * (1 + 3 * e / m); w /= w.mean()
    m, c = best
    # fp32 evaluation of GELU
    x = np.linspace(-7, 7, 700001).astype(np.float32)
    zz = np.minimum(np.abs(x) * np.float32(0.70710678), np.float32(zmax)).astype(np.float32)
    p = np.full_like(x, np.float32(c[-1]))
    for k in range(n - 2, -1, -1):
        p = (p * zz + np.float32(c[k])).astype(np.float32)
    a = (-(zz * p)).astype(np.float32)
    e2 = np.exp2(a.astype(np.float32)).astype(np.float32)
    erf_abs = (np.float32(1.0) - e2).astype(np.float32)
    er = np.copysign(erf_abs, x).astype(np.float32)
    h = (np.float32(0.5) * x).astype(np.float32)
    g = (h * er + h).astype(np.float32)
    gt = 0.5 * x.astype(np.float64) * (1 + serf(x.astype(np.float64) / np.sqrt(2)))
    print(f"n {n} zmax {zmax}: erfc fit abs err {m:.2e}; gelu abs err (fp32 eval) {np.abs(g - gt).max():.2e}; coeffs {[float(np.float32(v)) for v in c]}")
for n in (5, 6, 7):
    for zmax in (4.0, 4.3):
        run(n, zmax)
