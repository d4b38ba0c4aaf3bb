"""fp32 emulation of the r04 GELU epilogue (quartic P, no clamp): positivity / monotonicity of z P(z) and the error against scipy erf over several ranges."""
import numpy as np
from scipy.special import erf as serf
c = np.array([1.6278890371322632, 0.9185093641281128, 0.1486656814813614, -0.02959008701145649, 0.002944170031696558], dtype=np.float32)
z = np.linspace(0, 100, 1000001)
P = sum(float(ck) * z**k for k, ck in enumerate(c))
print("P min", P.min(), "at", z[P.argmin()], "; d(zP)/dz min", np.diff(z * P).min())
Q = np.array([1.1510913372039795, 0.4592546820640564, 0.052561257034540176, -0.007397521752864122, 0.0005204606568440795], dtype=np.float32)
def gelu_new(x):
    """csrc/common.hpp gelu_erf, operation by operation in fp32 (numpy rounds the product and the sum of an FMA separately: slightly pessimistic)"""
    x = x.astype(np.float32)
    a = np.abs(x)
    q = np.full_like(x, Q[4])
    for k in (3, 2, 1, 0):
        q = (q * a + Q[k]).astype(np.float32)
    t = (-(a * q) - np.float32(1.0)).astype(np.float32)
    with np.errstate(over="ignore", under="ignore"):
        e = np.exp2(t).astype(np.float32)
    return (np.maximum(x, np.float32(0)) - a * e).astype(np.float32)
for lo, hi in ((-8, 8), (-100, 100), (-1e4, 1e4)):
    x = np.linspace(lo, hi, 2000001)
    gt = 0.5 * x * (1 + serf(x / np.sqrt(2)))
    g = gelu_new(x)
    err = np.abs(g - gt)
    i = err.argmax()
    rel = (err / np.maximum(np.abs(gt), 1e-30))
    print(f"x in [{lo},{hi}]: max abs err {err.max():.3e} at x={x[i]:.4f} (g={gt[i]:.3e}); max rel err where |g|>1e-2: {rel[np.abs(gt)>1e-2].max():.2e}")
