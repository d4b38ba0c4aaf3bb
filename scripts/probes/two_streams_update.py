"""r06 probe: does the sequential memory update of S = 28 scenes run faster as TWO half-batches on two HIP streams (two decoder contexts, deep copy = own workspace)?
Every GEMM of a half-batch call covers half the CUs (126 tiles of 256 x 256), so two calls can be resident together, and the two chains drift out of phase: one
chain's HBM-bound work (fp32 residual epilogues, LayerNorm) next to the other's K loops -- the overlap that one block per CU never gets inside a launch."""
import copy, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from must3r_amd import synthetic as S
from must3r_amd.config import MUST3R_512
from test_model_gpu import build
Sn, V, H, W = 28, 20, 384, 512
enc, dec = build(MUST3R_512, "fp16wa")
dec2 = copy.deepcopy(dec)
imgs, ts = S.make_images(V, H, W, 0)
x1, pos1 = enc(imgs.cuda(), ts)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.stack([x1 + 0.05 * b * torch.randn(x1.shape, device="cuda", generator=g) for b in range(Sn)])       # [S, V, N, C]
pos = pos1.unsqueeze(0).expand(Sn, -1, -1, -1).contiguous()
tsb = ts.unsqueeze(0).expand(Sn, -1, -1)
mb = [2] + [1] * 18
def chain(d, xs, ps, tss, out):
    d.reserve_memory_tokens = V * 768
    mem, i = None, 0
    for nb in mb:
        mem, pm = d(xs[:, i:i + nb], ps[:, i:i + nb], tss[:, i:i + nb], mem, pointmaps_out=out[:, i:i + nb])
        i += nb
        yield
    return
def one():
    out = torch.empty((Sn, V, H, W, 7), device="cuda")
    for _ in chain(dec, x, pos, tsb, out): pass
    return out
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h = Sn // 2
def two(offset_calls=0):
    out = torch.empty((Sn, V, H, W, 7), device="cuda")
    torch.cuda.synchronize()
    ga = chain(dec, x[:h], pos[:h], tsb[:h], out[:h])
    gb = chain(dec2, x[h:], pos[h:], tsb[h:], out[h:])
    da = db = False
    k = 0
    while not (da and db):
        if not da:
            with torch.cuda.stream(s1):
                da = next(ga, "end") == "end"
        if not db and k >= offset_calls:
            with torch.cuda.stream(s2):
                db = next(gb, "end") == "end"
        k += 1
    s1.synchronize(); s2.synchronize()
    return out
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts_ = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts_.append(time.perf_counter() - t0)
    return min(ts_) * 1e3, sorted(ts_)[len(ts_) // 2] * 1e3
a = one(); b = two()
print("results equal within fp16 noise:", float((a - b).abs().max() / a.abs().max()))
for rep in range(2):
    print("one chain of 28 scenes        : min %.1f ms  median %.1f ms" % timed(one))
    print("two chains of 14, two streams : min %.1f ms  median %.1f ms" % timed(two))
    print("two chains, second one call late: min %.1f ms  median %.1f ms" % timed(lambda: two(1)))
