#!/usr/bin/env python
"""Per-stage kernel-class times of the bench step (S scenes in flight): HIP-event time per class inside encode / update / render and the
stage's wall time -- the difference is time no profiled kernel ran (launch gaps, host)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from must3r_amd import synthetic as S  # noqa: E402
from must3r_amd.config import MUST3R_512  # noqa: E402
from must3r_amd.engine import demo_mem_batches  # noqa: E402

Sn = int(os.environ.get("SCENES", "8"))
V, H, W = 20, 384, 512
dev = torch.device("cuda", 0)
enc, dec, _, _ = bench.build_models(MUST3R_512, os.environ.get("PRECISION", "fp16wa"), dev)
ts = S.make_images(V, H, W, seed=0)[1]
scenes = torch.stack([S.make_images(V, H, W, seed=1000 + b)[0] for b in range(Sn)]).to(dev)


def stages(profile):
    out = {}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def begin():
        for m in (enc, dec):
            m._context().set_profiling(profile)

    def end(name, a, b):
        torch.cuda.synchronize()
        prof = {}
        if profile:
            for m in (enc, dec):
                for k, v in m._context().get_profile().items():
                    p = prof.setdefault(k, [0.0, 0])
                    p[0] += v["ms"]; p[1] += v["calls"]
                m._context().set_profiling(False)
        out[name] = (a.elapsed_time(b), prof)

    begin(); ev[0].record()
    x, pos = enc(scenes.reshape(Sn * V, 3, H, W), ts.repeat(Sn, 1))
    ev[1].record(); end("encode", ev[0], ev[1])
    x, pos = x.view(Sn, V, *x.shape[1:]), pos.view(Sn, V, *pos.shape[1:])
    tsb = ts.unsqueeze(0).expand(Sn, -1, -1)
    begin(); ev[1].record()
    mem, i = None, 0
    for nb in demo_mem_batches(V):
        mem, _ = dec(x[:, i:i + nb], pos[:, i:i + nb], tsb[:, i:i + nb], mem)
        i += nb
    ev[2].record(); end("update", ev[1], ev[2])
    begin(); ev[2].record()
    dec(x, pos, tsb, mem, render=True)
    ev[3].record(); end("render", ev[2], ev[3])
    return out


stages(False)
plain = stages(False)
prof = stages(True)
for name in ("encode", "update", "render"):
    wall, _ = plain[name]
    wall_p, cls = prof[name]
    tot = sum(v[0] for v in cls.values())
    print(f"{name}: wall {wall:.2f} ms (profiled pass {wall_p:.2f}); kernels {tot:.2f} ms in {sum(v[1] for v in cls.values())} launches; "
          + ", ".join(f"{k} {v[0]:.2f}/{v[1]}" for k, v in sorted(cls.items(), key=lambda kv: -kv[1][0]) if v[1]))
