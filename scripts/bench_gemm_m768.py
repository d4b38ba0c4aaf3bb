"""The GEMM launches of a ONE-view memory update (M = 768) and of a one-view encoder call, per weight layout: time and output digest.
Run under M3R_BK128=0 / 1 / 2 to compare K-tile depths (digests must be equal: same accumulation order)."""
import hashlib, math, os, sys
import ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from must3r_amd import _lib as lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = torch.cuda.current_stream().cuda_stream
shapes = [("dec qkv", 768, 2304, 768, lib.EPI_STORE16, 2), ("dec proj", 768, 768, 768, lib.EPI_RESID_F32, 2), ("dec projq", 768, 768, 768, lib.EPI_STORE16, 2),
          ("dec fc1", 768, 3072, 768, lib.EPI_STORE16_GELU, 0), ("dec fc2", 768, 768, 3072, lib.EPI_RESID_F32, 0), ("dec kv", 768, 1536, 768, lib.EPI_STORE16, 2),
          ("dec fc1 w2", 768, 3072, 768, lib.EPI_STORE16_GELU, 2), ("dec fc2 w2", 768, 768, 3072, lib.EPI_RESID_F32, 2),
          ("enc qkv", 768, 3072, 1024, lib.EPI_STORE16, 2), ("enc proj", 768, 1024, 1024, lib.EPI_RESID_F32, 2), ("enc fc1", 768, 4096, 1024, lib.EPI_STORE16_GELU, 0),
          ("enc fc2", 768, 1024, 4096, lib.EPI_RESID_F32, 0)]
tot = 0.0
torch.manual_seed(0)
for name, M, N, K, epi, ws in shapes:
    A = torch.randn((M, K), device="cuda").half()
    W = (torch.randn((N, K * (2 if ws else 1)), device="cuda") / math.sqrt(K)).half()
    b = torch.randn((N,), device="cuda")
    out = torch.zeros((M, N), device="cuda", dtype=torch.float32 if epi == lib.EPI_RESID_F32 else torch.float16)
    def run():
        lib.check(L.must3r_hip_op_gemm(1, epi, P(A), P(W), P(b), P(out), M, N, K, K, N, None, None, 0, 0, None, 0, 0, 0, 0, 0, 0, ws, st))
    out.zero_(); run(); torch.cuda.synchronize()
    sha = hashlib.sha1(out.view(torch.int16 if out.element_size() == 2 else torch.int32).cpu().numpy().tobytes()).hexdigest()[:10]
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    tot += us
    print(f"  {name:11s} ws={ws} M={M:5d} N={N:5d} K={K:5d} {us:7.2f} us {2.0*M*N*K/us/1e6:7.1f} TF/s  sha {sha}")
print(f"M3R_BK128={os.environ.get('M3R_BK128', '(default)')}: total {tot:.1f} us")
