"""fp32 emulation of the r04 GELU epilogue (quartic P, no clamp): positivity / monotonicity of z P(z) and the error against scipy erf over several ranges."""
import numpy as np
from scipy.special import erf as serf
c = np.array([1.6278890371322632, 0.9185093641281128, 0.1486656814813614, -0.02959008701145649, 0.002944170031696558], dtype=np.float32)
z = np.linspace(0, 100, 1000001)
P = sum(float(ck) * z**k for k, ck in enumerate(c))
print("P min", P.min(), "at", z[P.argmin()], "; d(zP)/dz min", np.diff(z * P).min())
def gelu_new(x):
    x = x.astype(np.float32)
    ax = np.abs(x)
    zz = (ax * np.float32(0.70710678118654752440)).astype(np.float32)
    p = np.full_like(x, c[4])
    for k in (3, 2, 1, 0):
        p = (p * zz + c[k]).astype(np.float32)      # fma in hardware: single rounding; numpy double-rounds -> slightly pessimistic
    a = (-(zz * p)).astype(np.float32)
    with np.errstate(over="ignore", under="ignore"):
        e2 = np.exp2(a).astype(np.float32)
    t = (np.float32(0.5) * ax).astype(np.float32)
    return (np.maximum(x, np.float32(0)) - t * e2).astype(np.float32)
for lo, hi in ((-8, 8), (-100, 100), (-1e4, 1e4)):
    x = np.linspace(lo, hi, 2000001)
    gt = 0.5 * x * (1 + serf(x / np.sqrt(2)))
    g = gelu_new(x)
    err = np.abs(g - gt)
    i = err.argmax()
    rel = (err / np.maximum(np.abs(gt), 1e-30))
    print(f"x in [{lo},{hi}]: max abs err {err.max():.3e} at x={x[i]:.4f} (g={gt[i]:.3e}); max rel err where |g|>1e-2: {rel[np.abs(gt)>1e-2].max():.2e}")
