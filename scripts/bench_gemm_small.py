"""Microbenchmark of the update-pass (M = 768) GEMMs (r02; scripts/bench_gemm_m768.py is the r04 form with per-layout shapes and digests)."""
import os, sys, math, torch, ctypes as C
sys.path.insert(0, '/root/repo')
from must3r_amd import _lib as lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = torch.cuda.current_stream().cuda_stream
split = int(os.environ.get("SPLIT", "0"))
shapes = [("qkv", 768, 2304, 768, lib.EPI_STORE16), ("proj", 768, 768, 768, lib.EPI_RESID_F32), ("fc1", 768, 3072, 768, lib.EPI_STORE16_GELU),
          ("fc2", 768, 768, 3072, lib.EPI_RESID_F32), ("kv", 768, 1536, 768, lib.EPI_STORE16), ("init2 fc1", 1536, 3072, 768, lib.EPI_STORE16_GELU)]
if os.environ.get("STRIDE_EXP"):   # L2-channel camping check: same shapes with K off the power-of-two-ish strides
    shapes = [("fc2 K3072", 768, 768, 3072, lib.EPI_RESID_F32), ("fc2 K3136", 768, 768, 3136, lib.EPI_RESID_F32),
              ("fc2 K3008", 768, 768, 3008, lib.EPI_RESID_F32), ("proj K768", 768, 768, 768, lib.EPI_RESID_F32),
              ("proj K832", 768, 768, 832, lib.EPI_RESID_F32), ("proj K704", 768, 768, 704, lib.EPI_RESID_F32),
              ("fc1 K768", 768, 3072, 768, lib.EPI_STORE16_GELU), ("fc1 K832", 768, 3072, 832, lib.EPI_STORE16_GELU)]
tot = 0
torch.manual_seed(0)
for name, M, N, K, epi in shapes:
    A = torch.randn((M, K), device="cuda").half()
    W = (torch.randn((N, K * (2 if split else 1)), device="cuda") / math.sqrt(K)).half()
    b = torch.randn((N,), device="cuda")
    out = torch.zeros((M, N), device="cuda", dtype=torch.float32 if epi == lib.EPI_RESID_F32 else torch.float16)
    def run():
        lib.check(L.must3r_hip_op_gemm(1, epi, P(A), P(W), P(b), P(out), M, N, K, K, N, None, None, 0, 0, None, 0, 0, 0, 0, 0, 0, 2 if split else 0, st))
    if os.environ.get("M3R_CHECK"):   # digest of the output bits (compare kernels across env settings)
        import hashlib
        out.zero_(); run(); torch.cuda.synchronize()
        print("   sha", name, hashlib.sha1(out.view(torch.int16 if out.element_size() == 2 else torch.int32).cpu().numpy().tobytes()).hexdigest()[:12])
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    tot += us
    print(f"  {name:9s} M={M:5d} N={N:5d} K={K:5d} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s")
print(f"SPLIT={split} MIN_BIG={os.environ.get('M3R_GEMM_MIN_BIG','-')} MIN_BIG_SPLIT={os.environ.get('M3R_GEMM_MIN_BIG_SPLIT','-')}: total {tot:.1f} us")
