import os

import numpy as np
import torch

from conftest import GOLDEN


def rel_inf(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


# Tolerances of the HIP path against the reference / fp32 oracle, relative = ||d||_inf / ||ref||_inf.
#  fp16w2 (fp16 operands, split weights): the north-star tolerance, BASELINE.json "pointmaps within 1e-3 relative of
#          reference" (measured ~4e-4).
#  fp16   (single-pass fp16): measured 0.8-1.5e-3, i.e. AT the target but without margin -> asserted at 2e-3.
#  bf16   (what BASELINE.json's configs name): the reference's OWN bf16-autocast path is 1.05e-2 away from its fp32
#          path on this network; the HIP bf16 path (fp32 residual stream) must stay inside that
#          envelope (+10% for the max-norm's sampling noise; measured 6-11e-3).
#  fp16wa (fp16 operands, split weights everywhere but in the Mlp Linears): the north-star tolerance as well (emulated 6e-4,
#          scripts/emul/gemm_precision.py; measured: profiles/).
TOL = {"fp16w2": 1.0e-3, "fp16wa": 1.0e-3, "fp16": 2.0e-3, "bf16": 1.15e-2}
PRECISIONS = ("fp16w2", "fp16wa", "fp16", "bf16")
# the module default (what bench.py times) against the real-reference fixture of the benched scene, worst single view.
# r05: each view's error over the fixture's stored (every 8th) pixels is normalised by that view's max |value| over ALL its pixels
# and channels (`update_vmax` / `render_vmax`, written by oracle/make_golden.py from the full-resolution reference output) -- the normalisation of
# bench.py's all-pixel `parity_vs_cpu_oracle` figure (6.3-7.2e-4) -- instead of by the max of the sample (r04: 9.45e-4 against 1.0e-3, a 5 % margin that was
# an artefact of the sub-sampled range).  Asserted at 8e-4; measured figures in profiles/r05_test_metrics.jsonl.
# (in r04 units -- error / range of the SAMPLED pixels, which is 1.3-1.7x smaller than the full-resolution range -- this gate sits at ~1.3e-3, i.e. it is NOT
# tighter than r04's 1.0e-3; what makes it a real regression gate is its distance from the measured figures: sampled pixels 5.9-6.0e-4, and since r06 the reference's worst
# update / render views at FULL resolution -- tests/golden/must3r512_v20_fullviews.npz, asserted at the same 8e-4 -- 6.0-7.3e-4: profiles/r06_test_metrics.jsonl.)
TOL_DEFAULT_FIXTURE = 8.0e-4


def rel_inf_view(a_sub, ref_sub, vmax):
    """||a - ref||_inf over the stored pixels of one view / that view's full-resolution max |ref| (fixture key *_vmax)."""
    a, b = torch.as_tensor(a_sub).double(), torch.as_tensor(ref_sub).double()
    return ((a - b).abs().max() / max(float(vmax), 1e-30)).item()


from must3r_amd.synthetic import make_cam_pointmaps as cam_scene  # noqa: E402,F401
