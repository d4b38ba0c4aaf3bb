#!/usr/bin/env python
"""r06: persistent tile loop (must3r_hip_set_option("PERSIST", 0 / 1)) A/B on the chip-filling GEMM shapes of the S = 28 step, ONE process, interleaved
rounds (cdna_hip_programming.md section 5.4 rule 24), random operands, output bits compared (the loop changes nothing about the arithmetic).

    python scripts/r06_gemm_persist_ab.py            # prints one row per shape: us (min / median) per arm, TF/s, ratio, sha equality
"""
import ctypes as C
import hashlib
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from must3r_amd import _lib as lib  # noqa: E402

L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
st = torch.cuda.current_stream().cuda_stream
OPT = os.environ.get("OPT", "PERSIST")                         # the switch under test and its two values: OPT=G256_GM VALS=4,8
V0, V1 = (int(v) for v in os.environ.get("VALS", "0,1").split(","))
ROUNDS = int(os.environ.get("ROUNDS", "7"))
REPS = int(os.environ.get("REPS", "10"))

# (name, M, N, K, epilogue, split weights): encoder chunk of 40 views = 30720 rows; batched update step M = 28 x 768 = 21504; render M = 430080 is cut short here
SHAPES = [
    ("enc qkv  (sparse split)", 30720, 3072, 1024, lib.EPI_STORE16, 1),
    ("enc proj (sparse split)", 30720, 1024, 1024, lib.EPI_RESID_F32, 1),
    ("enc fc1  (plain, GELU)", 30720, 4096, 1024, lib.EPI_STORE16_GELU, 0),
    ("enc fc2  (plain, resid)", 30720, 1024, 4096, lib.EPI_RESID_F32, 0),
    ("upd qkv  (sparse split)", 21504, 2304, 768, lib.EPI_STORE16, 1),
    ("upd proj (sparse split)", 21504, 768, 768, lib.EPI_RESID_F32, 1),
    ("upd fc1  (plain, GELU)", 21504, 3072, 768, lib.EPI_STORE16_GELU, 0),
    ("upd fc2  (plain, resid)", 21504, 768, 3072, lib.EPI_RESID_F32, 0),
    ("upd K|V  (sparse split)", 21504, 1536, 768, lib.EPI_STORE16, 1),
    ("ren fc1  (plain, GELU)", 107520, 3072, 768, lib.EPI_STORE16_GELU, 0),
    ("ren fc2  (plain, resid)", 107520, 768, 3072, lib.EPI_RESID_F32, 0),
    ("ren proj (sparse split)", 107520, 768, 768, lib.EPI_RESID_F32, 1),
    ("yardstick 30720x4096x1024 store16", 30720, 4096, 1024, lib.EPI_STORE16, 0),
    ("yardstick 15360x3072x4096 store16", 15360, 3072, 4096, lib.EPI_STORE16, 0),
    ("4096^3 (one round)", 4096, 4096, 4096, lib.EPI_STORE16, 0),
    ("8192^3", 8192, 8192, 8192, lib.EPI_STORE16, 0),
]


def sha(t):
    return hashlib.sha1(t.view(torch.int16 if t.element_size() == 2 else torch.int32).cpu().numpy().tobytes()).hexdigest()[:12]


tot = {0: 0.0, 1: 0.0}
for name, M, N, K, epi, split in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K), device="cuda", generator=g).half()
    Wf = torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)
    b = torch.randn((N,), device="cuda", generator=g)
    f32out = epi in (lib.EPI_RESID_F32, lib.EPI_F32)
    out = torch.zeros((M, N), device="cuda", dtype=torch.float32 if f32out else torch.float16)
    if split:
        hi = Wf.half()
        W = torch.cat((hi, (Wf - hi.float()).half()), dim=1).contiguous()
        vals = torch.empty((K // 64, N, 32), device="cuda", dtype=torch.float16)
        idx = torch.empty((K // 64, N // 32, 64), device="cuda", dtype=torch.int32)
        lib.check(L.must3r_hip_op_sparse24_pack(P(Wf), N, K, P(vals), P(idx), st))

        def run():
            lib.check(L.must3r_hip_op_gemm_sp(epi, P(A), P(W), P(vals), P(idx), P(b), P(out), M, N, K, K, N, None, None, 0, 0, st))
    else:
        W = Wf.half()

        def run():
            lib.check(L.must3r_hip_op_gemm(1, epi, P(A), P(W), P(b), P(out), M, N, K, K, N, None, None, 0, 0, None, 0, 0, 0, 0, 0, 0, 0, st))
    # third arm: the vendor's plain product (hipBLASLt behind torch.matmul; no bias / epilogue: an upper bound for the fused launches), same box, same rounds
    vout = torch.empty((M, N), device="cuda", dtype=torch.float16)
    W16 = Wf.half()
    vendor = []
    digests, times = {}, {0: [], 1: []}
    for mode in (0, 1):
        lib.set_option(OPT, (V0, V1)[mode])
        out.zero_()
        run()
        torch.cuda.synchronize()
        digests[mode] = sha(out)
    for r in range(ROUNDS):
        for mode in (0, 1):
            lib.set_option(OPT, (V0, V1)[mode])
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                run()
            e1.record()
            torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) / REPS * 1e3)
        torch.matmul(A, W16.t(), out=vout)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            torch.matmul(A, W16.t(), out=vout)
        e1.record()
        torch.cuda.synchronize()
        vendor.append(e0.elapsed_time(e1) / REPS * 1e3)
    fl = 2.0 * M * N * K
    m0, m1 = statistics.median(times[0]), statistics.median(times[1])
    tot[0] += m0
    tot[1] += m1
    print(f"{name:36s} M={M:6d} N={N:5d} K={K:5d}  {OPT}={V0} {min(times[0]):8.1f} / {m0:8.1f} us ({fl / m0 / 1e6:7.1f} TF/s)   {OPT}={V1} {min(times[1]):8.1f} / {m1:8.1f} us "
          f"({fl / m1 / 1e6:7.1f} TF/s)   x{m0 / m1:5.3f}   bits {'equal' if digests[0] == digests[1] else 'DIFFER ' + digests[0] + ' ' + digests[1]}"
          f"   | vendor plain product {min(vendor):8.1f} / {statistics.median(vendor):8.1f} us ({fl / statistics.median(vendor) / 1e6:7.1f} TF/s)", flush=True)
    del A, W, Wf, out, vout, W16
print(f"sum of medians: {OPT}={V0} {tot[0]:.0f} us, {OPT}={V1} {tot[1]:.0f} us, x{tot[0] / tot[1]:.3f}")
lib.set_option(OPT, V0)
lib.set_option("PERSIST", 0)
