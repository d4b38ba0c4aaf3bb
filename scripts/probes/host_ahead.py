"""r06 probe: is the host ahead of the GPU during the one-scene-at-a-time memory update?  Per decoder call: the host time the call takes (no synchronisation), and the
queue depth in time (how long after the LAST call returned the GPU needs to drain).  A host that is ahead returns from each one-view call in ~0.4 ms (108 launches) and
leaves > 1 ms of GPU work queued per call; a host that blocks somewhere returns in the call's GPU time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from must3r_amd import synthetic as S
from must3r_amd.config import MUST3R_512
from test_model_gpu import build
enc, dec = build(MUST3R_512, "fp16wa")
imgs, ts = S.make_images(20, 384, 512, 0)
imgs = imgs.cuda()
x, pos = enc(imgs, ts)
def scene():
    mem, t_calls = None, []
    dec.reserve_memory_tokens = 20 * 768
    i = 0
    for nb in [2] + [1] * 18:
        t0 = time.perf_counter()
        mem, pm = dec(x[i:i + nb].unsqueeze(0), pos[i:i + nb].unsqueeze(0), ts[i:i + nb].unsqueeze(0), mem)
        t_calls.append(time.perf_counter() - t0)
        i += nb
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    return t_calls, time.perf_counter() - t0
for _ in range(3): scene()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    calls, drain = scene()
    tot = time.perf_counter() - t0
    print(f"update of one scene: {tot * 1e3:.2f} ms wall; host time per one-view call: min {min(calls[1:]) * 1e3:.3f} median {sorted(calls[1:])[9] * 1e3:.3f} max {max(calls[1:]) * 1e3:.3f} ms "
          f"(sum {sum(calls) * 1e3:.2f} ms); GPU work still queued when the last call returned: {drain * 1e3:.2f} ms")
