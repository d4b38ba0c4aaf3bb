#!/usr/bin/env python
"""Experiment: encoder/update overlap with CU-partitioned streams (engine.run_scene enc_cus / upd_cus / enc_chunk).

One process, one model; times the BASELINE scene under each configuration and checks the render is bit-identical to the
un-partitioned run (same kernels, same chunking => same bits).  Prints one JSON line per configuration.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from must3r_amd.config import MUST3R_512  # noqa: E402
from must3r_amd import synthetic as S  # noqa: E402
from must3r_amd.engine import run_scene  # noqa: E402


def main():
    torch.set_num_threads(16)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    prec = os.environ.get("PREC", "fp16w2")
    enc, dec, _, _ = bench.build_models(MUST3R_512, prec, device)
    H, W, V = 384, 512, 20
    imgs, ts = S.make_images(V, H, W, seed=0)
    imgs, ts = imgs.to(device), ts.to(device)
    steps = int(os.environ.get("STEPS", "5"))

    def timed(**kw):
        for _ in range(2):
            out = run_scene(enc, dec, imgs, ts, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = run_scene(enc, dec, imgs, ts, **kw)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, out["render"].clone()

    configs = [dict(overlap=False), dict(overlap=True)]
    if os.environ.get("EXP") == "chunks":      # encoder chunking on the second stream only (no CU masks)
        configs += [dict(overlap=True, enc_chunk=c) for c in (3, 9, 18)]
        masks = []
    else:
        masks = (6, 9, 18)
    for chunk in masks:
        for enc_cus in (64, 96, 128, 160, 192):
            configs.append(dict(overlap=True, enc_cus=enc_cus, enc_chunk=chunk))
    for enc_cus, upd_cus in (((128, 128), (96, 160), (160, 96), (64, 192), (128, 256), (192, 64)) if masks else ()):
        configs.append(dict(overlap=True, enc_cus=enc_cus, upd_cus=upd_cus, enc_chunk=6))
    ref = None
    for kw in configs:
        try:
            ms, ren = timed(**kw)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"config": kw, "error": repr(e)[:300]}), flush=True)
            continue
        if ref is None:
            ref = ren
        same = bool(torch.equal(ren, ref))
        err = float((ren - ref).abs().max() / ref.abs().max())
        print(json.dumps({"config": kw, "ms_per_scene": round(ms, 2), "views_per_s": round(V / ms * 1e3, 1),
                          "bit_identical_to_first": same, "rel_diff": err}), flush=True)


if __name__ == "__main__":
    main()
