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
# the module default (what bench.py times) against the real-reference fixture of the benched scene, worst single view: tightened with the
# default's measured margin (tests/test_zz_r04_gpu.py::test_benched_configuration_scenes_in_flight_vs_reference_fixture).  r04, fp16wa, 8 scenes in
# flight: worst update view 9.45e-4, worst render view 9.29e-4 (one scene at a time: 8.6e-4) -- deterministic (no atomics on the path), so the thin
# margin does not flake; DESIGN.md section 4 says what it would cost to widen it.
TOL_DEFAULT_FIXTURE = 1.0e-3


from must3r_amd.synthetic import make_cam_pointmaps as cam_scene  # noqa: E402,F401
