#!/usr/bin/env python
"""Same-node yardstick: the vendor kernels on the shapes of the bench step, on the box the hand kernels run on.

NOT part of the product: nothing under must3r_amd/ or bench.py's timed region imports this file.  It answers one question -- what do
hipBLASLt (behind torch.matmul) and the SDPA back-end reach on THESE shapes, on random data, on this box -- so that the hand kernels are
held against a measured ceiling instead of a model.

* GEMM: out[M,N] = A[M,K] . W[N,K]^T, fp16 / bf16 operands, fp32 accumulate; the chip-filling shapes of scripts/exp_gemm256.py (M = 15360) and the
  encoder-chunk shapes of the S = 20 step (M = 30720).  Plain product only: the vendor call has no GELU / RoPE / fp32 read-modify-write epilogue,
  so these are UPPER bounds for the fused launches (the hand kernels' epilogues cost 10-25 us per round of tiles on top of the product).
* Attention: F.scaled_dot_product_attention on the render cross-attention (768 queries x 15360 keys per view, 12 heads, d = 64) and the encoder
  self-attention (768 x 768, 16 heads) shapes.
Prints one line per shape: average us over ITERS back-to-back launches (HIP events), TF/s algorithmic.
"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ITERS = int(os.environ.get("ITERS", "20"))


def timeit(fn, iters=ITERS, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    print(f"torch {torch.__version__}  device {torch.cuda.get_device_name(0)}  iters {ITERS}", flush=True)
    shapes = [("enc qkv", 15360, 3072, 1024), ("enc proj", 15360, 1024, 1024), ("enc fc1", 15360, 4096, 1024), ("enc fc2", 15360, 1024, 4096),
              ("dec qkv", 15360, 2304, 768), ("dec proj", 15360, 768, 768), ("dec fc1", 15360, 3072, 768), ("dec fc2", 15360, 768, 3072),
              ("dec kv", 15360, 1536, 768),
              ("enc40 qkv", 30720, 3072, 1024), ("enc40 proj", 30720, 1024, 1024), ("enc40 fc1", 30720, 4096, 1024), ("enc40 fc2", 30720, 1024, 4096),
              ("k4096", 15360, 3072, 4096), ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)]
    for dt in (torch.float16, torch.bfloat16):
        tot_t = tot_f = 0.0
        for name, M, N, K in shapes:
            g = torch.Generator(device="cuda").manual_seed(M + N + K)
            A = torch.randn((M, K), device="cuda", generator=g).to(dt)
            W = (torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)).to(dt)
            out = torch.empty((M, N), device="cuda", dtype=dt)
            us = timeit(lambda: torch.matmul(A, W.t(), out=out))
            fl = 2.0 * M * N * K
            if M == 15360 and name != "k4096":
                tot_t += us
                tot_f += fl
            print(f"gemm {str(dt)[6:]:8s} {name:10s} M={M:6d} N={N:5d} K={K:5d} {us:8.1f} us {fl / us / 1e6:7.1f} TF/s", flush=True)
        print(f"gemm {str(dt)[6:]:8s} nine shapes (M = 15360): {tot_t:.0f} us, {tot_f / tot_t / 1e6:.1f} TF/s", flush=True)
        # fused-epilogue equivalents the vendor path would need as separate passes (what nn.Linear + GELU / residual add cost the reference)
        M, N, K = 15360, 4096, 1024
        A = torch.randn((M, K), device="cuda").to(dt)
        W = (torch.randn((N, K), device="cuda") / math.sqrt(K)).to(dt)
        b = torch.randn((N,), device="cuda").to(dt)
        us = timeit(lambda: F.gelu(F.linear(A, W, b)))
        print(f"gemm {str(dt)[6:]:8s} enc fc1 + bias + GELU (two kernels) {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)

    # attention: (batch, heads, q, k, d)
    att = [("render CA 20 views", 20, 12, 768, 15360, 64), ("render CA 1 view", 1, 12, 768, 15360, 64), ("encoder SA 40 views", 40, 16, 768, 768, 64),
           ("decoder SA 20 views", 20, 12, 768, 768, 64)]
    for dt in (torch.float16, torch.bfloat16):
        for name, B, H, nq, nk, d in att:
            q = torch.randn((B, H, nq, d), device="cuda").to(dt)
            # the memory is shared by the views of a scene in the model; SDPA wants it per batch entry: expand (no copy) keeps the bytes honest
            k = torch.randn((1, H, nk, d), device="cuda").to(dt).expand(B, H, nk, d) if nk > nq else torch.randn((B, H, nk, d), device="cuda").to(dt)
            v = torch.randn((1, H, nk, d), device="cuda").to(dt).expand(B, H, nk, d) if nk > nq else torch.randn((B, H, nk, d), device="cuda").to(dt)
            for backend_name in ("default",):
                try:
                    us = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
                    fl = 4.0 * B * H * nq * nk * d
                    print(f"sdpa {str(dt)[6:]:8s} {name:22s} B={B:3d} H={H:2d} q={nq:5d} k={nk:6d} {us:9.1f} us {fl / us / 1e6:7.1f} TF/s", flush=True)
                except Exception as e:   # a back-end that refuses a shape is a result too
                    print(f"sdpa {str(dt)[6:]:8s} {name:22s} failed: {type(e).__name__}: {str(e)[:120]}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
