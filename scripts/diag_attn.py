import sys, torch, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from must3r_amd import _lib as lib
from test_ops_gpu import attn_ref, P, stream
def run(views, heads, Rq, Rk, dt=1):
    tdt = torch.float16
    g = torch.Generator(device="cuda").manual_seed(11)
    D = heads*64
    q = (torch.randn((Rq, D), device="cuda", generator=g) * 1.5).to(tdt)
    kvm = (torch.randn((Rk, 2 * D), device="cuda", generator=g) * 1.5).to(tdt)
    k, v = kvm[:, :D], kvm[:, D:]
    o = torch.full((Rq, D), float("nan"), device="cuda", dtype=tdt)
    tab = torch.tensor(views, dtype=torch.int32, device="cuda")
    lib.check(lib.load().must3r_hip_op_attention(dt, P(q), P(k), P(v), P(o), q.stride(0), k.stride(0), v.stride(0), o.stride(0), heads, P(tab), len(views), max(vw[1] for vw in views), stream()))
    torch.cuda.synchronize()
    ref = attn_ref(q.cpu(), k.cpu(), v.cpu(), views, heads)
    d = (o.cpu().double() - ref).abs()
    for vi, vw in enumerate(views):
        for h in range(heads):
            blk = d[vw[0]:vw[0]+vw[1], h*64:(h+1)*64]
            print("  view", vi, vw, "head", h, "maxerr %.4f" % blk.max().item(), "bad rows", (blk.max(dim=1).values > 0.02).nonzero().flatten().tolist()[:12])
for name, (views, heads, Rq, Rk) in {
  "one view partial end": ([(0,100,0,1000,300,496)], 1, 100, 1000),
  "one view end in tile, start aligned": ([(0,64,0,640,128,200)], 1, 64, 640),
  "skip from start partial end": ([(0,70,0,392,0,196)], 1, 70, 392),
  "second half": ([(0,70,0,392,196,392)], 1, 70, 392),
  "two views": ([(0,70,0,392,0,196),(70,70,0,392,196,392)], 1, 140, 392),
  "skip inside one tile": ([(0,64,0,256,70,100)], 1, 64, 256),
}.items():
    print(name); run(views, heads, Rq, Rk)
