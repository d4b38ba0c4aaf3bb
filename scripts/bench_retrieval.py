#!/usr/bin/env python
"""Timing of the retrieval front-end on encoder tokens (prewhiten + projector + attention + postwhiten + top-300 / pooling)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from must3r_amd import synthetic as S  # noqa: E402
from must3r_amd.retrieval import RetrievalModel  # noqa: E402


class _B(torch.nn.Module):
    embed_dim = 1024


sd = S.make_retrieval_state_dict(1024, seed=0)
m = RetrievalModel(_B(), prewhiten=-1, postwhiten=-1, hdims=[1024], nfeat=300).cuda().eval()
m.load_state_dict(sd, strict=True)
for Bn in (1, 20):
    x = torch.randn((Bn, 768, 1024), device="cuda")
    for name, fn in (("forward_local", m.forward_local), ("forward_global", m.forward_global)):
        for _ in range(3):
            fn(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            fn(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 10 * 1e3
        fl = Bn * 768 * 1024 * 1024 * 2 * 3
        print(f"{name:15s} images {Bn:3d} x 768 tokens x 1024: {ms:7.3f} ms  ({fl / ms / 1e9:6.2f} TFLOP/s over the three 1024^2 products, two of them float64)",
              flush=True)
