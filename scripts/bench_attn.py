"""Microbenchmark of the attention kernels on the scene's shapes.  default = attn3_kernel; M3R_ATTN=4 on an experiment build (make EXTRA=-DM3R_ATTN_EXPERIMENTS) = the 16-bit 32x32-tile attn4_kernel;
FP8=1 times the e4m3 Q/K variant (MX-scaled 32x32x64 MFMA for Q K^T).  Prints median / min over interleaved rounds."""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from must3r_amd import _lib as lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = torch.cuda.current_stream().cuda_stream
FP8 = int(os.environ.get("FP8", "0"))
def make(name, heads, nviews, nq, nk, self_attn, nsplit=0):
    D = heads * 64
    dt = torch.float8_e4m3fn if FP8 else torch.float16
    torch.manual_seed(0)
    if self_attn:
        qkv = (torch.randn((nviews * nq, 3 * D), device="cuda")).half()
        q, k, v = qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:]
        if FP8:   # e4m3 q | k rows, V stays fp16
            qk8 = qkv[:, :2 * D].to(dt)
            q, k = qk8[:, :D], qk8[:, D:]
        views = [(i * nq, nq, i * nq, nq, 0, 0) for i in range(nviews)]
    else:
        q = torch.randn((nviews * nq, D), device="cuda").to(dt)
        if FP8:   # memory rows [K e4m3 (D bytes) | V fp16 (2 D bytes)]
            rows = torch.empty((nk, 3 * D), dtype=torch.uint8, device="cuda")
            k, v = rows[:, :D].view(dt), rows[:, D:].view(torch.float16)
            k.copy_(torch.randn((nk, D), device="cuda").to(dt)); v.copy_(torch.randn((nk, D), device="cuda").half())
        else:
            kv = torch.randn((nk, 2 * D), device="cuda").to(dt)
            k, v = kv[:, :D], kv[:, D:]
        views = [(i * nq, nq, 0, nk, 0, 0) for i in range(nviews)]
    o = torch.empty((nviews * nq, D), device="cuda", dtype=torch.float16)
    tab = torch.tensor(views, dtype=torch.int32, device="cuda")
    ws = None
    if nsplit > 1:
        ws = torch.empty((L.must3r_hip_attention_scratch_bytes(nsplit, nviews * nq, heads),), dtype=torch.uint8, device="cuda")
    code = lib.F16 | (lib.ATTN_FP8 if FP8 else 0)
    def go():
        lib.check(L.must3r_hip_op_attention(code, P(q), P(k), P(v), P(o), q.stride(0), k.stride(0), v.stride(0), o.stride(0), heads, P(tab),
                                            len(views), nq, nsplit, P(ws), nviews * nq if nsplit > 1 else 0, st))
    return name, go, 4.0 * nviews * nq * nk * D, (q, k, v, o, tab, ws)
cases = [make("render CA 20v nk15360", 12, 20, 768, 15360, False), make("enc SA 20v n768", 16, 20, 768, 768, True),
         make("update CA 4v nk7680 (S=4)", 12, 4, 768, 7680, False), make("update CA 4v nk7680 s3 (S=4)", 12, 4, 768, 7680, False, 3),
         make("update CA 1v nk7680 s7", 12, 1, 768, 7680, False, 7), make("update CA 1v nk14592 s8", 12, 1, 768, 14592, False, 8),
         make("update SA 1v n768", 12, 1, 768, 768, True), make("render CA 20v nk1960 (224)", 12, 10, 196, 1960, False),
         make("update CA 1v nk7680 s10", 12, 1, 768, 7680, False, 10), make("update CA 1v nk7680 s14", 12, 1, 768, 7680, False, 14),
         make("update CA 1v nk14592 s14", 12, 1, 768, 14592, False, 14), make("render CA 20v nk15360 s2", 12, 20, 768, 15360, False, 2)]
if os.environ.get("FIXED_COST"):   # fixed cost of a split launch: 1, 2, 6, 12 tiles per block at s = 10
    cases = [make(f"update CA nk{nk} s10", 12, 1, 768, nk, False, 10) for nk in (640, 1280, 3840, 7680, 15360)] + \
            [make(f"update CA nk{nk} s1", 12, 1, 768, nk, False, 0) for nk in (64, 640)]
if os.environ.get("SELF_SWEEP"):
    # r06: where does an encoder self-attention launch (768 keys = 12 tiles per block, 800 TF/s in the step) spend its time?  Same query blocks, the
    # number of key tiles swept: the slope is the loop's cost per tile, the intercept what a block pays before / after its loop (Q fragments, first tile's DMA latency,
    # normalise + store) plus the launch; 8 views = 768 blocks = ONE round of the 3 blocks a CU holds, 40 views = five rounds (the encoder chunk of the step).
    def make_sweep(nviews, nk):
        D, nq, heads = 1024, 768, 16
        torch.manual_seed(0)
        q = torch.randn((nviews * nq, D), device="cuda").half()
        kv = torch.randn((nviews * nk, 2 * D), device="cuda").half()
        k, v = kv[:, :D], kv[:, D:]
        o = torch.empty((nviews * nq, D), device="cuda", dtype=torch.float16)
        tab = torch.tensor([(i * nq, nq, i * nk, nk, 0, 0) for i in range(nviews)], dtype=torch.int32, device="cuda")
        def go():
            lib.check(L.must3r_hip_op_attention(lib.F16, P(q), P(k), P(v), P(o), q.stride(0), k.stride(0), v.stride(0), o.stride(0), heads, P(tab), nviews, nq, 0, None, 0, st))
        return f"SA {nviews:2d}v x 16 heads, 768 q, {nk:4d} keys ({nk // 64:2d} tiles)", go, 4.0 * nviews * nq * nk * D, (q, kv, o, tab)
    cases = [make_sweep(nv, nk) for nv in (8, 40) for nk in (64, 128, 256, 384, 768, 1536, 3072)]
for name, go, fl, _ in cases:
    go()
torch.cuda.synchronize()
res = {c[0]: [] for c in cases}
for rnd in range(5):
    for name, go, fl, _ in cases:
        iters = 5 if fl > 5e10 else 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): go()
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / iters)
print(f"M3R_ATTN={os.environ.get('M3R_ATTN','2')} QF={os.environ.get('M3R_ATTN_QF','1')} FP8={FP8}")
for name, go, fl, _ in cases:
    r = sorted(res[name]); med, mn = r[len(r)//2], r[0]
    print(f"  {name:28s} median {med*1e3:9.1f} us {fl/med/1e9:8.1f} TF/s   min {mn*1e3:9.1f} us {fl/mn/1e9:8.1f} TF/s", flush=True)

if os.environ.get("SELF_SWEEP"):
    import numpy as np
    for nv in (8, 40):
        rows = [(int(c[0].split("(")[1].split()[0]), sorted(res[c[0]])[len(res[c[0]]) // 2] * 1e3) for c in cases if c[0].startswith(f"SA {nv:2d}v")]
        t, us = np.array([r[0] for r in rows], float), np.array([r[1] for r in rows], float)
        slope, icpt = np.polyfit(t[2:], us[2:], 1)      # (4 tiles and up: the one- and two-tile launches sit on the launch floor)
        blocks = nv * 16 * 6
        rounds = blocks / 768.0
        print(f"  {nv} views: {blocks} blocks = {rounds:.1f} rounds of 768; per launch {slope:.2f} us per key tile + {icpt:.1f} us; per ROUND {slope / rounds:.3f} us per tile + {icpt / rounds:.2f} us fixed"
              f"  ->  at 12 tiles the fixed part is {icpt / (icpt + 12 * slope) * 100:.0f} % of the launch; loop alone = {4.0 * nv * 768 * 64 * 1024 / slope / 1e6:.0f} TF/s")
