"""TEST INFRASTRUCTURE ONLY.  CPU baseline leg of bench.py: times the oracle (a port of the reference's CPU path,
``kind: "port"``) on a bounded sample of the benchmark workload and stores its pointmaps for the parity numbers.

    python oracle/cpu_baseline.py --views 20 --H 384 --W 512 --threads 32 --check-fixture --out /tmp/x.npz

``--views 20`` is the metric's own workload (BASELINE.json configs[2]: 20-view 384x512 scene, schedule [2,1,...,1]); with
``--check-fixture`` the oracle's pointmaps are compared with ``tests/golden/must3r512_v20.npz`` (outputs of the REAL
reference on the same seeded inputs, oracle/make_golden.py model_big), so the oracle is pinned at the headline
configuration on every bench run.

Runs in its own process so bench.py can bound it with a timeout and so that the CPU thread pool does not
interfere with the GPU process.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--H", type=int, default=384)
    ap.add_argument("--W", type=int, default=512)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", required=True)
    ap.add_argument("--check-fixture", action="store_true")
    a = ap.parse_args()
    from oracle import must3r_ref as R
    from must3r_amd.config import MUST3R_512
    from must3r_amd import synthetic as S
    torch.set_num_threads(a.threads)
    cfg = MUST3R_512
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(20, a.H, a.W, a.seed)   # same images as bench.py rank 0; the sample is the first `views`
    imgs, ts = imgs[:a.views], ts[:a.views]
    tm = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        upd, ren, _ = R.run_scene(sde, sdd, cfg, imgs, ts, timings=tm)
    dt = time.perf_counter() - t0
    np.savez(a.out, render=ren.numpy(), update=upd.numpy())
    info = {"seconds": dt, "views": a.views, "threads": torch.get_num_threads(), "stages_s": tm}
    fx = os.path.join(ROOT, "tests", "golden", "must3r512_v20.npz")
    if a.check_fixture and a.views == 20 and (a.H, a.W, a.seed) == (384, 512, 0) and os.path.exists(fx):
        g = np.load(fx)
        ps = int(g["meta"][3])
        rel = lambda x, r: float(np.abs(x - r).max() / np.abs(r).max())  # noqa: E731
        info["oracle_vs_reference_fixture"] = {"update_rel_inf": rel(upd[:, ::ps, ::ps].numpy(), g["update"]),
                                               "render_rel_inf": rel(ren[:, ::ps, ::ps].numpy(), g["render"]),
                                               "fixture": "tests/golden/must3r512_v20.npz (real reference, 20 views)"}
    print(json.dumps(info))


if __name__ == "__main__":
    main()
