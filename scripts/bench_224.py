"""configs[1] alone: MUSt3R_224, 10 views of 224x224 -- wall time per scene, and (under rocprofv3 --kernel-trace) how much of it is kernels."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from must3r_amd.config import MUST3R_224
from must3r_amd import synthetic as S
from must3r_amd.engine import run_scene
import bench
enc, dec, _, _ = bench.build_models(MUST3R_224, "fp16w2", torch.device("cuda"))
imgs, ts = S.make_images(10, 224, 224, seed=0)
imgs, ts = imgs.cuda(), ts.cuda()
for _ in range(3): run_scene(enc, dec, imgs, ts)
torch.cuda.synchronize()
n = int(os.environ.get("N", "20"))
t0 = time.perf_counter()
for _ in range(n): run_scene(enc, dec, imgs, ts)
t1 = time.perf_counter()           # host done queueing
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"224x224 10-view scene: {(t2 - t0) / n * 1e3:.2f} ms wall per scene ({10 * n / (t2 - t0):.1f} views/s); host queueing finished after {(t1 - t0) / n * 1e3:.2f} ms per scene")
