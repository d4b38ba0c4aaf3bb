#!/usr/bin/env python
"""8-wave 256-tile GEMM (M3R_GEMM256=2) against the 4-wave kernels (M3R_GEMM256=0): prints, per shape and epilogue, the time
and a checksum of the output bits.  Run once per mode and diff the checksums (the kernels must agree bit for bit)."""
import ctypes as C
import hashlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from must3r_amd import _lib as lib  # noqa: E402

L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
st = torch.cuda.current_stream().cuda_stream
split = int(os.environ.get("SPLIT", "0"))
plain16 = int(os.environ.get("PLAIN16", "0"))   # fp16 operands with plain weights (the Mlp Linears of MUST3R_F16_WA)
tdt, dti = (torch.float16, 1) if (split or plain16) else (torch.bfloat16, 0)
shapes = [("enc qkv", 15360, 3072, 1024, lib.EPI_STORE16), ("enc proj", 15360, 1024, 1024, lib.EPI_RESID_F32),
          ("enc fc1", 15360, 4096, 1024, lib.EPI_STORE16_GELU), ("enc fc2", 15360, 1024, 4096, lib.EPI_RESID_F32),
          ("dec qkv", 15360, 2304, 768, lib.EPI_STORE16), ("dec proj", 15360, 768, 768, lib.EPI_RESID_F32),
          ("dec fc1", 15360, 3072, 768, lib.EPI_STORE16_GELU), ("dec fc2", 15360, 768, 3072, lib.EPI_RESID_F32),
          ("dec kv", 15360, 1536, 768, lib.EPI_STORE16), ("tail M", 15360 - 100, 1024, 1024, lib.EPI_F32),
          ("enc18 fc1", 13824, 4096, 1024, lib.EPI_STORE16_GELU), ("k192", 4096, 1024, 192, lib.EPI_STORE16)]
if os.environ.get("ONLY") or os.environ.get("EXTRA"):   # timing-experiment shapes (the fixed per-tile cost; a long K loop)
    shapes += [("k64", 15360, 3072, 64, lib.EPI_STORE16), ("k128", 15360, 3072, 128, lib.EPI_STORE16), ("k4096", 15360, 3072, 4096, lib.EPI_STORE16),
               ("k64 f32", 15360, 3072, 64, lib.EPI_RESID_F32), ("k16384", 15360, 1024, 16384, lib.EPI_STORE16)]
if os.environ.get("MROWS"):   # the decoder shapes at another row count (S scenes in flight: M = S * 768 in the batched update)
    mr = int(os.environ["MROWS"])
    shapes = [(n, mr, N, K, e) for n, M, N, K, e in shapes if n.startswith("dec")]
tot_t = tot_f = 0.0
only = os.environ.get("ONLY")   # comma-separated shape names
for name, M, N, K, epi in shapes:
    if only and name not in only.split(","):
        continue
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K), device="cuda", generator=g).to(tdt)
    Wf = torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)
    if split:
        hi = Wf.half()
        W = torch.cat((hi, (Wf - hi.float()).half()), dim=1).contiguous()
    else:
        W = Wf.to(tdt)
    b = torch.randn((N,), device="cuda", generator=g)
    f32out = epi in (lib.EPI_RESID_F32, lib.EPI_F32)
    out = torch.zeros((M, N), device="cuda", dtype=torch.float32 if f32out else tdt)

    def run():
        lib.check(L.must3r_hip_op_gemm(dti, epi, P(A), P(W), P(b), P(out), M, N, K, K, N, None, None, 0, 0, None, 0, 0, 0, 0, 0, 0,
                                       2 if split else 0, st))
    run()
    torch.cuda.synchronize()
    digest = hashlib.sha1(out.view(torch.int16 if out.element_size() == 2 else torch.int32).cpu().numpy().tobytes()).hexdigest()[:12]
    ref = A[:256].double() @ Wf.double().t() + b.double()
    if epi == lib.EPI_STORE16_GELU:
        ref = torch.nn.functional.gelu(ref)
    err = ((out[:256].double() - ref).abs().max() / ref.abs().max()).item()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    if True:
        if epi == lib.EPI_RESID_F32:   # the in-place residual epilogue accumulates: start from the same zeros again
            out.zero_()
            run()
            torch.cuda.synchronize()
        digest2 = hashlib.sha1(out.view(torch.int16 if out.element_size() == 2 else torch.int32).cpu().numpy().tobytes()).hexdigest()[:12]
        if digest2 != digest:
            print(f"RACE: {name}: output bits changed between launches ({digest} -> {digest2})", flush=True)
    fl = 2.0 * M * N * K
    if M >= 15000 or only or os.environ.get("MROWS"):
        tot_t += ms
        tot_f += fl
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d} {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s  err {err:.2e}  sha {digest}", flush=True)
print(f"mode {os.environ.get('M3R_GEMM256', '1')} G256S {os.environ.get('M3R_G256S', '0')} split {split} plain16 {plain16}: big shapes {tot_t * 1e3:.0f} us, {tot_f / tot_t / 1e9:.1f} TF/s (algorithmic)")
