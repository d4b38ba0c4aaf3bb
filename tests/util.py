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


# tolerances of the HIP path against the fp32 oracle / reference, relative = ||d||_inf / ||ref||_inf.
#  fp16 operands: the north-star tolerance (BASELINE.json: "pointmaps within 1e-3 relative of reference").
#  bf16 operands: the reference's OWN bf16-autocast path is 1.05e-2 away from its fp32 path on this
#  network; the HIP bf16 path (fp32 residual stream) must stay inside that envelope.
TOL = {"fp16": 1.0e-3, "bf16": 1.05e-2}
