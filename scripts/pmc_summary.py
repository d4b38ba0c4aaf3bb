#!/usr/bin/env python
"""gpurun_out/pmc_{FETCH,WRITE}_SIZE.json -> profiles/rNN_pmc_traffic.json (per-kernel HBM traffic per launch).

Corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half the bytes of a wide coalesced streaming read (16 B/lane global_load and buffer_load...lds alike),
so it is doubled; WRITE_SIZE is taken as is (uncalibrated).  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import json
import re
import subprocess
import sys

out = sys.argv[1] if len(sys.argv) > 1 else "profiles/r01_pmc_traffic.json"
f = json.load(open("gpurun_out/pmc_FETCH_SIZE.json"))["kernels"]
w = json.load(open("gpurun_out/pmc_WRITE_SIZE.json"))["kernels"]
res = {}
for k in sorted(set(f) | set(w)):
    fk, wk = f.get(k, {}), w.get(k, {})
    if not (k.startswith("_ZN3m3r") or k.startswith("m3r::")):
        continue
    fetch, write = fk.get("mean_per_launch", 0.0), wk.get("mean_per_launch", 0.0)
    short = re.sub(r"^_ZN3m3r\d+", "", k)
    res[k] = {"launches": fk.get("launches", wk.get("launches")), "FETCH_SIZE_KiB_mean": round(fetch, 1),
              "WRITE_SIZE_KiB_mean": round(write, 1), "hbm_bytes_per_launch_corrected": int((2 * fetch + write) * 1024)}
import os
try:   # the commit the counters were taken at (gpurun ships no .git: scripts/gpu_pmc.sh passes it through M3R_COMMIT)
    import os
    commit = os.environ.get("M3R_COMMIT") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"
except Exception:
    commit = "?"
json.dump({"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python bench.py --gpus 1 --steps 1 --warmup 1 "
                      "--no-cpu-baseline --no-alt --no-configs --no-single  (two separate passes)",
           "commit": commit, "scenes": int(os.environ.get("M3R_PMC_SCENES", "20")),
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE halves wide coalesced reads; WRITE_SIZE uncalibrated)",
           "kernels": res}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"] * (kv[1]["launches"] or 0))[:8]:
    print(k[:80], v)
