#!/usr/bin/env python
"""Per-shape timing of the split-weight chip-filling GEMMs: dense two-pass (M3R_SPARSE_LO has no say here: the op entry is explicit) against the
2:4-sparse low part (must3r_hip_op_gemm_sp), per epilogue.  Shapes: the split launches of the S = 28 step."""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from must3r_amd import _lib as lib  # noqa: E402

L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
st = torch.cuda.current_stream().cuda_stream
shapes = [("enc qkv", 30720, 3072, 1024, lib.EPI_QKV_ROPE), ("enc qkv (store)", 30720, 3072, 1024, lib.EPI_STORE16), ("enc proj", 30720, 1024, 1024, lib.EPI_RESID_F32),
          ("dec qkv", 21504, 2304, 768, lib.EPI_QKV_ROPE), ("dec qkv (store)", 21504, 2304, 768, lib.EPI_STORE16), ("dec proj", 21504, 768, 768, lib.EPI_RESID_F32),
          ("dec kv", 21504, 1536, 768, lib.EPI_STORE16), ("dec projq", 21504, 768, 768, lib.EPI_STORE16)]
npos = 64
buf = (C.c_float * (npos * 32))()
L.must3r_hip_rope_table(100.0, 1.0, npos, buf)
tab = torch.tensor(list(buf), device="cuda")
for name, M, N, K, epi in shapes:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K), device="cuda", generator=g).half()
    Wf = torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)
    hi = Wf.half()
    W2 = torch.cat((hi, (Wf - hi.float()).half()), dim=1).contiguous()
    vals = torch.empty((K // 64, N, 32), device="cuda", dtype=torch.float16)
    idx = torch.empty((K // 64, N // 32, 64), device="cuda", dtype=torch.int32)
    lib.check(L.must3r_hip_op_sparse24_pack(P(Wf.contiguous()), N, K, P(vals), P(idx), st))
    b = torch.randn((N,), device="cuda", generator=g)
    pos = torch.stack((torch.arange(M, device="cuda") // 32 % 24, torch.arange(M, device="cuda") % 32), -1).contiguous()
    f32out = epi in (lib.EPI_RESID_F32, lib.EPI_F32)
    out = torch.zeros((M, N), device="cuda", dtype=torch.float32 if f32out else torch.float16)
    rope = epi == lib.EPI_QKV_ROPE
    rc = (N // 3 * 2) if rope else 0

    def dense():
        lib.check(L.must3r_hip_op_gemm(1, epi, P(A), P(W2), P(b), P(out), M, N, K, K, N, P(pos) if rope else None, P(tab) if rope else None, rc, npos if rope else 0,
                                       None, 0, 0, 0, 0, 0, 0, 2, st))

    def sparse():
        lib.check(L.must3r_hip_op_gemm_sp(epi, P(A), P(W2), P(vals), P(idx), P(b), P(out), M, N, K, K, N, P(pos) if rope else None, P(tab) if rope else None, rc,
                                          npos if rope else 0, st))
    res = []
    for fn in (dense, sparse):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * M * N * K
    print(f"{name:18s} M={M:6d} N={N:5d} K={K:5d} epi {epi}: dense {res[0]:8.1f} us ({fl / res[0] / 1e6:6.1f} TF/s)   sparse {res[1]:8.1f} us ({fl / res[1] / 1e6:6.1f} TF/s)   x{res[0] / res[1]:.3f}", flush=True)
