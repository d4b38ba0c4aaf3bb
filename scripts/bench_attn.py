"""Microbenchmark of the attention kernel on the scene's big shapes (render cross attention, encoder self attention)."""
import sys, torch, ctypes as C
sys.path.insert(0, '/root/repo')
from must3r_amd import _lib as lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = torch.cuda.current_stream().cuda_stream
def run(name, heads, nviews, nq, nk, self_attn, iters=5):
    D = heads * 64
    if self_attn:
        qkv = (torch.randn((nviews * nq, 3 * D), device="cuda")).bfloat16()
        q, k, v = qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:]
        views = [(i * nq, nq, i * nq, nq, 0, 0) for i in range(nviews)]
    else:
        q = torch.randn((nviews * nq, D), device="cuda").bfloat16()
        kv = torch.randn((nk, 2 * D), device="cuda").bfloat16()
        k, v = kv[:, :D], kv[:, D:]
        views = [(i * nq, nq, 0, nk, 0, 0) for i in range(nviews)]
    o = torch.empty((nviews * nq, D), device="cuda", dtype=torch.bfloat16)
    tab = torch.tensor(views, dtype=torch.int32, device="cuda")
    def go():
        lib.check(L.must3r_hip_op_attention(0, P(q), P(k), P(v), P(o), q.stride(0), k.stride(0), v.stride(0), o.stride(0), heads, P(tab),
                                            len(views), nq, 0, None, 0, st))
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 4.0 * nviews * nq * nk * D
    print(f"{name:22s} {ms*1e3:9.1f} us {fl/ms/1e9:8.1f} TF/s", flush=True)
run("render CA 20v nk15360", 12, 20, 768, 15360, False)
run("enc SA 20v n768", 16, 20, 768, 768, True)
run("update CA 1v nk7680", 12, 1, 768, 7680, False)
