#!/usr/bin/env python
"""Timing of postprocess with / without compute_cam (HBM-bound row of DESIGN.md section 3)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from must3r_amd import synthetic as S  # noqa: E402
from must3r_amd.engine import postprocess  # noqa: E402

pm20 = S.make_cam_pointmaps(20, 384, 512, focal=450.0, noise=0.02, seed=1).cuda()
for V in (20, 1, 4):
    pm = pm20[:V].contiguous()     # slices of one resident tensor: freeing big tensors inside the loop stalls the next timings
    for cc in (False, True):
        for _ in range(3):
            postprocess(pm, compute_cam=cc)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(30):
            postprocess(pm, compute_cam=cc)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 30 * 1e3
        print(f"views {V:3d} compute_cam {cc!s:5s} {ms * 1e3:8.1f} us  {pm.numel() // 7 * 56 / ms / 1e6:8.1f} GB/s", flush=True)
