#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 scripts/probes/build/pipe_rates 2>&1 | tee gpurun_out/pipe_rates_stride.txt
python - <<'P'
import os, sys, math, torch, ctypes as C
sys.path.insert(0, os.getcwd())
from must3r_amd import _lib as lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = torch.cuda.current_stream().cuda_stream
for name, M, N, K, epi in [("fc1 K768", 768, 3072, 768, lib.EPI_STORE16_GELU), ("fc1 K1536", 768, 3072, 1536, lib.EPI_STORE16_GELU), ("fc1 K3072", 768, 3072, 3072, lib.EPI_STORE16_GELU),
                           ("st16 K768", 768, 3072, 768, lib.EPI_STORE16), ("st16 K3072", 768, 3072, 3072, lib.EPI_STORE16)]:
    A = torch.randn((M, K), device="cuda").half(); W = (torch.randn((N, 2 * K), device="cuda") / math.sqrt(K)).half(); b = torch.randn((N,), device="cuda")
    out = torch.zeros((M, N), device="cuda", dtype=torch.float16)
    run = lambda: lib.check(L.must3r_hip_op_gemm(1, epi, P(A), P(W), P(b), P(out), M, N, K, K, N, None, None, 0, 0, None, 0, 0, 0, 0, 0, 0, 2, st))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(f"  {name:12s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
P
