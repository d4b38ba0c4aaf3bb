"""Microbenchmark of the GEMM kernel on the big-batch shapes of the 20-view scene (encoder and render pass)."""
import os, sys, time, math, torch, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from must3r_amd import _lib as lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
shapes = [("enc qkv", 15360, 3072, 1024, lib.EPI_STORE16), ("enc proj", 15360, 1024, 1024, lib.EPI_RESID_F32),
          ("enc fc1", 15360, 4096, 1024, lib.EPI_STORE16_GELU), ("enc fc2", 15360, 1024, 4096, lib.EPI_RESID_F32),
          ("dec qkv", 15360, 2304, 768, lib.EPI_STORE16), ("dec proj", 15360, 768, 768, lib.EPI_RESID_F32),
          ("dec fc1", 15360, 3072, 768, lib.EPI_STORE16_GELU), ("dec fc2", 15360, 768, 3072, lib.EPI_RESID_F32),
          ("enc6 fc1", 4608, 4096, 1024, lib.EPI_STORE16_GELU)]
st = torch.cuda.current_stream().cuda_stream
tot_t = tot_f = 0
split = int(os.environ.get("SPLIT", "0"))
tdt, dti = (torch.float16, 1) if split else (torch.bfloat16, 0)
for name, M, N, K, epi in shapes:
    A = torch.randn((M, K), device="cuda").to(tdt)
    Wf = torch.randn((N, K), device="cuda") / math.sqrt(K)
    if split:
        hi = Wf.half(); W = torch.cat((hi, (Wf - hi.float()).half()), dim=1).contiguous()
    else:
        W = Wf.to(tdt)
    b = torch.randn((N,), device="cuda")
    out = torch.zeros((M, N), device="cuda", dtype=torch.float32 if epi == lib.EPI_RESID_F32 else tdt)
    def run():
        lib.check(L.must3r_hip_op_gemm(dti, epi, P(A), P(W), P(b), P(out), M, N, K, K, N, None, None, 0, 0, None, 0, 0, 0, 0, 0, 0, 2 if split else 0, st))
    if epi == lib.EPI_STORE16:   # correctness of the variant under test
        run(); torch.cuda.synchronize()
        ref = A[:512].double() @ (Wf if split else W.float()).double().t() + b.double()
        err = ((out[:512].double() - ref).abs().max() / ref.abs().max()).item()
        assert err < 2e-2 if not split else err < 2e-3, (name, err)
        print(f"  [{name} rel err {err:.2e}]", end="")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * M * N * K
    tot_t += ms; tot_f += fl
    print(f"  {name:9s} M={M:6d} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")
print(f"variant {os.environ.get('M3R_GEMM_BIG','0')}: total {tot_t*1e3:.0f} us, {tot_f/tot_t/1e9:.1f} TF/s")
