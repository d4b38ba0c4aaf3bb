#!/usr/bin/env python
"""Timing of the brute-force nearest-neighbour query (SLAM keyframe test): one 384x512 frame subsampled by 4 / 2
(12 k / 49 k queries) against 10..100 keyframes of 12 k points."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from must3r_amd.slam_nn import nn_distances  # noqa: E402

for nq, nd in ((12288, 122880), (12288, 1228800), (49152, 1228800), (49152, 4915200)):
    db = (torch.randn((nd, 3), device="cuda") * 3).contiguous()
    q = (torch.randn((nq, 3), device="cuda") * 3).contiguous()
    for _ in range(2):
        nn_distances(db, q)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        nn_distances(db, q)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    pairs = nq * nd
    print(f"queries {nq:6d} database {nd:8d}: {ms:8.3f} ms  {pairs / ms / 1e9:7.2f} Tpairs/s  {8 * pairs / ms / 1e9:7.1f} TFLOP/s fp32 (8 flop/pair; peak 157)", flush=True)
