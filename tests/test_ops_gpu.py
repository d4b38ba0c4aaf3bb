"""GPU (-m gpu): every HIP kernel alone, through the C ABI, against an fp64 torch evaluation of the same
16-bit-rounded operands (so only accumulation order and output rounding differ)."""
import ctypes as C
import json
import math
import os

import pytest
import torch

from conftest import ROOT
from util import rel_inf

pytestmark = pytest.mark.gpu

DT = {"bf16": (0, torch.bfloat16, 2.0 ** -8), "fp16": (1, torch.float16, 2.0 ** -11)}  # id, torch dtype, unit round-off


def record(name, **kw):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "test_metrics.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **kw}) + "\n")


@pytest.fixture(scope="module")
def lib():
    from must3r_amd import _lib
    return _lib


def stream():
    return torch.cuda.current_stream().cuda_stream


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def run_gemm(lib, dt, epi, A, W, bias, out, ldc=None, pos=None, tab=None, rope_cols=0, npos=0, bias2=None, row2=0, acc=0,
             ntok=0, gw=0, H=0, Wimg=0, wsplit=0):
    M, K = A.shape
    N = W.shape[0]
    L = lib.load()
    lib.check(L.must3r_hip_op_gemm(DT[dt][0], epi, P(A), P(W), P(bias), P(out), M, N, K, A.stride(0),
                                   ldc if ldc is not None else N, P(pos), P(tab), rope_cols, npos, P(bias2), row2, acc,
                                   ntok, gw, H, Wimg, wsplit, stream()))
    torch.cuda.synchronize()


def test_transposing_lds_read_mapping(lib):
    """ds_read_b64_tr_b16: inside a 16-lane group the 16 8-byte chunks form a 4x16 row-major matrix and lane i gets
    column i (common.hpp lds_read_tr4).  The attention kernel's V^T operand depends on exactly this."""
    out = torch.full((256,), -1, dtype=torch.int16, device="cuda")
    lib.check(lib.load().must3r_hip_debug_tr_probe(P(out), stream()))
    torch.cuda.synchronize()
    got = out.cpu().view(64, 4).tolist()
    record("tr_probe", mapping=got)
    exp = [[(l // 16) * 64 + (l % 16) + 16 * e for e in range(4)] for l in range(64)]
    assert got == exp, got


SHAPES = [(200, 128, 64), (768, 768, 768), (1000, 1024, 256), (3072, 1024, 512), (3000, 2304, 768), (196, 3072, 1024),
          (12, 128, 128), (15260, 1024, 128)]   # the last one fills the chip: 256-row tiles (gemm256k / gemm256), ragged last row block


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_store_gelu_resid_f32(lib, dt, shape):
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    tdt, u = DT[dt][1], DT[dt][2]
    A = (torch.randn((M, K), device="cuda", generator=g)).to(tdt)
    W = (torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)).to(tdt)
    bias = torch.randn((N,), device="cuda", generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    # 16-bit store
    out = torch.empty((M, N), device="cuda", dtype=tdt)
    run_gemm(lib, dt, lib.EPI_STORE16, A, W, bias, out)
    e1 = rel_inf(out, ref)
    assert torch.allclose(out.double(), ref, rtol=2 * u, atol=2 * u), e1
    # asymmetric check: transposed result must NOT match (catches row/col swaps on square shapes)
    if M == N:
        assert not torch.allclose(out.double().t(), ref, rtol=2 * u, atol=2 * u)
    # GELU
    run_gemm(lib, dt, lib.EPI_STORE16_GELU, A, W, bias, out)
    refg = torch.nn.functional.gelu(ref)
    e2 = rel_inf(out, refg)
    assert torch.allclose(out.double(), refg, rtol=2 * u, atol=2 * u), e2
    # residual: x += A W^T + b
    x0 = torch.randn((M, N), device="cuda", generator=g)
    x = x0.clone()
    run_gemm(lib, dt, lib.EPI_RESID_F32, A, W, bias, x)
    e3 = rel_inf(x, x0.double() + ref)
    assert torch.allclose(x.double(), x0.double() + ref, rtol=1e-5, atol=1e-4), e3
    # fp32 out + second bias on rows >= row2, then accumulate
    b2 = torch.randn((N,), device="cuda", generator=g)
    row2 = M // 3
    o32 = torch.empty((M, N), device="cuda")
    run_gemm(lib, dt, lib.EPI_F32, A, W, bias, o32, bias2=b2, row2=row2)
    ref2 = ref.clone()
    ref2[row2:] += b2.double()
    e4 = rel_inf(o32, ref2)
    assert torch.allclose(o32.double(), ref2, rtol=1e-5, atol=1e-4), e4
    run_gemm(lib, dt, lib.EPI_F32, A, W, bias, o32, acc=1)
    assert torch.allclose(o32.double(), ref2 + (ref - bias.double()), rtol=1e-5, atol=2e-4)
    record("gemm", dt=dt, shape=shape, store=e1, gelu=e2, resid=e3, f32=e4)


@pytest.mark.parametrize("shape", [(768, 768, 768), (3072, 1024, 512), (200, 128, 64), (15260, 1024, 128)])
def test_gemm_split_weights(lib, shape):
    """W = W_hi + W_lo in fp16, two MFMA passes: the weight rounding term must vanish (error -> activation rounding
    + fp32 accumulation only) and the result must equal the explicit two-GEMM sum."""
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(21)
    A = torch.randn((M, K), device="cuda", generator=g).half()
    Wf = torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)
    hi = Wf.half()
    lo = (Wf - hi.float()).half()
    W2 = torch.cat((hi, lo), dim=1).contiguous()
    bias = torch.randn((N,), device="cuda", generator=g)
    out = torch.empty((M, N), device="cuda")
    L = lib.load()
    lib.check(L.must3r_hip_op_gemm(1, lib.EPI_F32, P(A), P(W2), P(bias), P(out), M, N, K, K, N, None, None, 0, 0, None, 0, 0,
                                   0, 0, 0, 0, 2, stream()))
    torch.cuda.synchronize()
    ref = A.double() @ Wf.double().t() + bias.double()
    plain = A.double() @ hi.double().t() + bias.double()
    e_split, e_plain = rel_inf(out, ref), rel_inf(plain, ref)
    record("gemm_split", shape=shape, split=e_split, plain_fp16_weights=e_plain)
    assert e_split < 2e-6 and e_split < e_plain / 20, (e_split, e_plain)


def _split_w(Wf):
    hi = Wf.half()
    return torch.cat((hi, (Wf - hi.float()).half()), dim=1).contiguous()


@pytest.mark.parametrize("shape", [(768, 3072, 768), (768, 2304, 768), (700, 2304, 768), (1536, 1536, 768)])
def test_gemm96_tiles_same_bits_as_small_tiles(lib, shape):
    """The one-view update's N >= 1536 launches (M = 768: qkv, fc1, feedback fc1) run on 96 x 96 tiles (gemm96_kernel).  Same
    accumulation order as every other tile shape: the rows of a full launch must be bit-identical to the same rows computed by a
    64-row launch (which the dispatcher sends to the 64 x 64 kernels), for the store, GELU and residual epilogues, and inside the
    fp64 tolerance."""
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = torch.randn((M, K), device="cuda", generator=g).half()
    Wf = torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)
    W2 = _split_w(Wf)
    bias = torch.randn((N,), device="cuda", generator=g)
    ref = A.double() @ Wf.double().t() + bias.double()
    L = lib.load()

    def go(epi, a, out):
        lib.check(L.must3r_hip_op_gemm(1, epi, P(a), P(W2), P(bias), P(out), a.shape[0], N, K, K, N, None, None, 0, 0, None, 0, 0,
                                       0, 0, 0, 0, 2, stream()))
        torch.cuda.synchronize()
    rows = [0, 64, (M // 64 - 1) * 64]
    for epi, odt in ((lib.EPI_STORE16, torch.float16), (lib.EPI_STORE16_GELU, torch.float16), (lib.EPI_F32, torch.float32)):
        full = torch.empty((M, N), device="cuda", dtype=odt)
        go(epi, A, full)
        want = torch.nn.functional.gelu(ref) if epi == lib.EPI_STORE16_GELU else ref
        tol = 2 * 2.0 ** -11 if odt == torch.float16 else 1e-5
        assert torch.allclose(full.double(), want, rtol=tol, atol=tol * 4), rel_inf(full, want)
        for r0 in rows:
            part = torch.empty((64, N), device="cuda", dtype=odt)
            go(epi, A[r0:r0 + 64], part)
            assert torch.equal(part, full[r0:r0 + 64]), (epi, r0)
    record("gemm96", shape=shape, err_f32=rel_inf(full, ref))


@pytest.mark.parametrize("ws,N", [(0, 1024), (2, 1024), (2, 1536)])
def test_gemm256_tiles_same_bits_as_small_tiles(lib, ws, N):
    """Chip-filling launches run on 256-row tiles (plain weights: gemm256p_kernel -- r05, phase-staggered 64-deep K-tiles; gemm256k_kernel before --;
    split weights: gemm256_kernel, or gemm256p_kernel's 256 x 128 form where that fills its rounds better: N = 1536) with the
    batched epilogue (operands of four row fragments loaded in front of their stores, 16-byte full-line stores after two cross-lane
    exchanges).  Every row must carry the bits of the same row computed by a 64-row launch (64 x 64 tiles, per-fragment epilogue
    path for the ragged fragment), for all four epilogues, including the ragged last row block (M = 15260: 156 rows, last fragment 12).
    K = 192 = three K-tiles: the steady, the second-to-last and the last form of the K loop."""
    M, K = 15260, 192
    g = torch.Generator(device="cuda").manual_seed(77 + ws)
    A = torch.randn((M, K), device="cuda", generator=g).half()
    Wf = torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)
    W = _split_w(Wf) if ws == 2 else Wf.half().contiguous()
    bias = torch.randn((N,), device="cuda", generator=g)
    x0 = torch.randn((M, N), device="cuda", generator=g)
    ref = A.double() @ (Wf.double() if ws == 2 else W.double()).t() + bias.double()
    L = lib.load()

    def go(epi, a, out):
        lib.check(L.must3r_hip_op_gemm(1, epi, P(a), P(W), P(bias), P(out), a.shape[0], N, K, K, N, None, None, 0, 0, None, 0, 0,
                                       0, 0, 0, 0, ws, stream()))
        torch.cuda.synchronize()
    starts = [0, 64 * 117, 15104, 15232]   # 15232: the last 28 rows (one full fragment + the ragged one)
    for epi, odt in ((lib.EPI_STORE16, torch.float16), (lib.EPI_STORE16_GELU, torch.float16), (lib.EPI_F32, torch.float32),
                     (lib.EPI_RESID_F32, torch.float32)):
        full = x0.clone() if epi == lib.EPI_RESID_F32 else torch.full((M, N), 7.0, device="cuda", dtype=odt)
        go(epi, A, full)
        want = torch.nn.functional.gelu(ref) if epi == lib.EPI_STORE16_GELU else (ref + x0.double() if epi == lib.EPI_RESID_F32 else ref)
        tol = 2 * 2.0 ** -11 if odt == torch.float16 else 1e-5
        assert torch.allclose(full.double(), want, rtol=tol, atol=tol * 4), (epi, rel_inf(full, want))
        for r0 in starts:
            r1 = min(r0 + 64, M)
            part = x0[r0:r1].clone() if epi == lib.EPI_RESID_F32 else torch.empty((r1 - r0, N), device="cuda", dtype=odt)
            go(epi, A[r0:r1], part)
            assert torch.equal(part, full[r0:r1]), (epi, r0)
    record("gemm256_bits", ws=ws, N=N)



def _prune24(lo):
    """keep the 2 entries of largest magnitude of every 4 consecutive k (ties: the lower k) -- misc.hip::sparse24_pack_kernel"""
    g = lo.reshape(lo.shape[0], -1, 4)
    mag = g.abs().float()
    # stable ranking: larger magnitude first, lower index first among equals
    key = mag * 8 - torch.arange(4, device=lo.device, dtype=torch.float32) * 1e-30
    order = torch.argsort(-mag, dim=-1, stable=True)[..., :2]
    mask = torch.zeros_like(g, dtype=torch.bool).scatter_(-1, order, True)
    return (g * mask).reshape(lo.shape)


@pytest.mark.parametrize("shape", [(15360, 1536, 768), (15260, 768, 192), (21504, 2304, 768), (15360, 1024, 1024)])
def test_gemm_sparse_low_part(lib, shape):
    """r05: the chip-filling split-weight kernel with a 2:4-SPARSE low part (gemm256p_kernel<.., WS = 3>: one v_smfmac_f32_16x16x64_f16 per output
    fragment and 64-deep K-tile for the low product).  Reference: the product with exactly the operands the kernel multiplies, A . (W_hi + P(W_lo))^T in
    fp64 -- P keeps the 2 largest |.| of every 4 consecutive k -- for the store, RoPE-less, fp32 and residual epilogues, ragged last row block and
    grouped K-tile tails included; and the distance of that product from the exact one stays at the weight-rounding level the emulation predicted."""
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K), device="cuda", generator=g).half()
    Wf = torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)
    hi = Wf.half()
    lo = (Wf - hi.float()).half()
    W2 = torch.cat((hi, lo), dim=1).contiguous()
    bias = torch.randn((N,), device="cuda", generator=g)
    x0 = torch.randn((M, N), device="cuda", generator=g)
    L = lib.load()
    vals = torch.empty((K // 64, N, 32), device="cuda", dtype=torch.float16)
    idx = torch.empty((K // 64, N // 32, 64), device="cuda", dtype=torch.int32)
    lib.check(L.must3r_hip_op_sparse24_pack(P(Wf.contiguous()), N, K, P(vals), P(idx), stream()))
    torch.cuda.synchronize()
    # the packed values are the kept entries in k order: vals[t][n][8 g + 2 q + {0, 1}] for k = 64 t + 16 g + 4 q + ...
    lo_p = _prune24(lo)
    kept = lo_p.reshape(N, K // 64, 4, 4, 4)                       # [n][t][g][q][e]
    nz = torch.argsort((kept == 0).to(torch.int8), dim=-1, stable=True)[..., :2]          # positions of the two kept entries (zeros last), in k order
    want_vals = torch.gather(kept, -1, nz.sort(dim=-1).values).reshape(N, K // 64, 32).permute(1, 0, 2)
    # (a kept entry that is itself 0.0 makes "which zero" ambiguous: compare values only where the row group has two non-zeros)
    two = ((kept != 0).sum(-1) == 2).reshape(N, K // 64, 16).permute(1, 0, 2).repeat_interleave(2, dim=-1)
    assert torch.equal(vals[two], want_vals[two])
    ref = A.double() @ (hi.double() + lo_p.double()).t() + bias.double()
    exact = A[:512].double() @ Wf[:, :].double().t() + bias.double()

    def go(epi, out):
        lib.check(L.must3r_hip_op_gemm_sp(epi, P(A), P(W2), P(vals), P(idx), P(bias), P(out), M, N, K, K, N, None, None, 0, 0, stream()))
        torch.cuda.synchronize()
    for epi, odt in ((lib.EPI_STORE16, torch.float16), (lib.EPI_F32, torch.float32), (lib.EPI_RESID_F32, torch.float32)):
        out = x0.clone() if epi == lib.EPI_RESID_F32 else torch.full((M, N), 7.0, device="cuda", dtype=odt)
        go(epi, out)
        want = ref + x0.double() if epi == lib.EPI_RESID_F32 else ref
        tol = 2 * 2.0 ** -11 if odt == torch.float16 else 2e-6
        assert torch.allclose(out.double(), want, rtol=tol, atol=tol * 4), (epi, rel_inf(out, want))
        if epi == lib.EPI_F32:
            e_exact = rel_inf(out[:512], exact)
            record("gemm_sparse_lo", shape=shape, err_vs_operands=rel_inf(out, want), err_vs_exact=e_exact)
            assert e_exact < 1.5e-4, e_exact   # weight rounding left after the sparse low part: ~0.45 x 2^-12 per weight, averaged over K


# (every shape launches the 256-row kernels over MORE than one round of 256 tiles: 484 / 432 / 472 / 472 tiles; ragged last row block; K-tile counts 16 / 12 / 3 / 1)
@pytest.mark.parametrize("shape", [(30720 + 70, 1024, 1024), (9000, 3072, 768), (30000, 1024, 192), (30000, 1024, 64)])
def test_gemm_persistent_tile_loop_same_bits(lib, shape):
    """r06: the chip-filling kernels walk their tiles in a PERSISTENT loop (one block per CU; gemm256p_kernel / gemm256s_kernel <.., PERS = 1>) whenever a launch
    has more tiles than CUs -- set_option("PERSIST", 0) launches one block per tile as before.  The loop changes nothing about the arithmetic: the outputs of the two
    forms are bit-identical for plain and sparse-split weights, every epilogue, ragged last row blocks, several rounds, and the persistent form leaves the same bits
    launch after launch (no stale LDS across a tile boundary)."""
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K), device="cuda", generator=g).half()
    Wf = torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)
    hi = Wf.half()
    W2 = torch.cat((hi, (Wf - hi.float()).half()), dim=1).contiguous()
    bias = torch.randn((N,), device="cuda", generator=g)
    x0 = torch.randn((M, N), device="cuda", generator=g)
    L = lib.load()
    vals = torch.empty((K // 64, N, 32), device="cuda", dtype=torch.float16)
    idx = torch.empty((K // 64, N // 32, 64), device="cuda", dtype=torch.int32)
    lib.check(L.must3r_hip_op_sparse24_pack(P(Wf.contiguous()), N, K, P(vals), P(idx), stream()))
    ref = A[:300].double() @ hi[:, :].double().t() + bias.double()

    def go(epi, out, sparse):
        if sparse:
            lib.check(L.must3r_hip_op_gemm_sp(epi, P(A), P(W2), P(vals), P(idx), P(bias), P(out), M, N, K, K, N, None, None, 0, 0, stream()))
        else:
            lib.check(L.must3r_hip_op_gemm(1, epi, P(A), P(hi), P(bias), P(out), M, N, K, K, N, None, None, 0, 0, None, 0, 0, 0, 0, 0, 0, 0, stream()))
        torch.cuda.synchronize()
    try:
        for sparse in (False, True):
            for epi, odt in ((lib.EPI_STORE16, torch.float16), (lib.EPI_STORE16_GELU, torch.float16), (lib.EPI_F32, torch.float32), (lib.EPI_RESID_F32, torch.float32)):
                outs = []
                for mode in (0, 1, 1):
                    lib.set_option("PERSIST", mode)
                    out = x0.clone() if epi == lib.EPI_RESID_F32 else torch.full((M, N), 7.0, device="cuda", dtype=odt)
                    go(epi, out, sparse)
                    outs.append(out)
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), (sparse, epi)
                if epi == lib.EPI_F32 and not sparse:   # (and the bits are those of the product)
                    assert torch.allclose(outs[1][:300].double(), ref, rtol=2e-6, atol=1e-5)
    finally:
        lib.set_option("PERSIST", 0)


def test_gemm_sparse_tile_widths_same_bits(lib):
    """gemm256s_kernel (256 x 256 sparse tiles; its position dwords come in through inline-asm loads the compiler does not track) against
    gemm256p_kernel<.., WS = 3> (256 x 128 tiles, positions through LDS): same operands, same accumulation order -- identical bits, on a shape both fill."""
    M, N, K = 15360, 3072, 1024
    g = torch.Generator(device="cuda").manual_seed(77)
    A = torch.randn((M, K), device="cuda", generator=g).half()
    Wf = torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)
    hi = Wf.half()
    W2 = torch.cat((hi, (Wf - hi.float()).half()), dim=1).contiguous()
    bias = torch.randn((N,), device="cuda", generator=g)
    L = lib.load()
    vals = torch.empty((K // 64, N, 32), device="cuda", dtype=torch.float16)
    idx = torch.empty((K // 64, N // 32, 64), device="cuda", dtype=torch.int32)
    lib.check(L.must3r_hip_op_sparse24_pack(P(Wf.contiguous()), N, K, P(vals), P(idx), stream()))
    outs = {}
    try:
        for w in (1, 0):
            lib.set_option("SPARSE_256", w)
            for rep in range(3):
                out = torch.zeros((M, N), device="cuda", dtype=torch.float16)
                lib.check(L.must3r_hip_op_gemm_sp(lib.EPI_STORE16, P(A), P(W2), P(vals), P(idx), P(bias), P(out), M, N, K, K, N, None, None, 0, 0, stream()))
                torch.cuda.synchronize()
                assert w not in outs or torch.equal(outs[w], out), ("launch-to-launch", w, rep)
                outs[w] = out
    finally:
        lib.set_option("SPARSE_256", 1)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("geom", [(2, 14, 14, 128), (1, 24, 32, 768), (3, 3, 4, 1024), (4, 24, 32, 256), (20, 24, 32, 512)])
def test_gemm_qkv_rope(lib, dt, geom):
    from oracle import must3r_ref as R
    V, gh, gw, Cdim = geom
    N = gh * gw
    M = V * N
    Hn = Cdim // 64
    g = torch.Generator(device="cuda").manual_seed(5)
    tdt, u = DT[dt][1], DT[dt][2]
    A = torch.randn((M, Cdim), device="cuda", generator=g).to(tdt)
    W = (torch.randn((3 * Cdim, Cdim), device="cuda", generator=g) / math.sqrt(Cdim)).to(tdt)
    bias = torch.randn((3 * Cdim,), device="cuda", generator=g) * 0.1
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    pos = torch.stack((ys.reshape(-1), xs.reshape(-1)), -1).view(1, N, 2).expand(V, -1, -1).contiguous()
    npos = 64
    buf = (C.c_float * (npos * 32))()
    lib.load().must3r_hip_rope_table(100.0, 1.0, npos, buf)
    tab = torch.tensor(list(buf), device="cuda")
    out = torch.empty((M, 3 * Cdim), device="cuda", dtype=tdt)
    run_gemm(lib, dt, lib.EPI_QKV_ROPE, A, W, bias, out, pos=pos.cuda().view(M, 2), tab=tab, rope_cols=2 * Cdim, npos=npos)
    lin = (A.double() @ W.double().t() + bias.double()).cpu().view(V, N, 3, Hn, 64)
    q, k, v = (lin[:, :, i].permute(0, 2, 1, 3).float() for i in range(3))
    q = R.rope2d(q, pos)
    k = R.rope2d(k, pos)
    ref = torch.stack((q, k, v), dim=2).permute(0, 3, 2, 1, 4).reshape(M, 3 * Cdim)  # [V,N,3,H,64]
    e = rel_inf(out.cpu(), ref)
    record("gemm_rope", dt=dt, geom=geom, err=e)
    assert torch.allclose(out.cpu().double(), ref.double(), rtol=2 * u, atol=4 * u), e


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_gemm_head_pixel_shuffle(lib, dt):
    nv, gh, gw, D = 2, 3, 4, 128
    H, Wd, N = gh * 16, gw * 16, gh * gw
    g = torch.Generator(device="cuda").manual_seed(9)
    tdt = DT[dt][1]
    A = torch.randn((nv * N, D), device="cuda", generator=g).to(tdt)
    W = (torch.randn((1792, D), device="cuda", generator=g) / math.sqrt(D)).to(tdt)   # reference row order c*256+i*16+j
    b = torch.randn((1792,), device="cuda", generator=g)
    f = (A.double() @ W.double().t() + b.double()).view(nv, gh, gw, 7, 16, 16).permute(0, 1, 4, 2, 5, 3).reshape(nv, H, Wd, 7)
    idx = torch.arange(1792, device="cuda")
    old = (idx % 7) * 256 + idx // 7          # new row (i*16+j)*7+c <- old row c*256 + i*16 + j
    Wp, bp = W[old].contiguous(), b[old].contiguous()
    out = torch.full((nv, H, Wd, 7), float("nan"), device="cuda")
    run_gemm(lib, dt, lib.EPI_HEAD, A, Wp, bp, out, ldc=0, ntok=N, gw=gw, H=H, Wimg=Wd)
    e = rel_inf(out, f)
    record("gemm_head", dt=dt, err=e)
    assert torch.allclose(out.double(), f, rtol=1e-5, atol=1e-4), e
    run_gemm(lib, dt, lib.EPI_HEAD, A, Wp, bp, out, ldc=0, acc=1, ntok=N, gw=gw, H=H, Wimg=Wd)
    assert torch.allclose(out.double(), 2 * f - b.double()[old].view(16, 16, 7)[None, None, :, None].expand(nv, gh, 16, gw, 16, 7)
                          .reshape(nv, H, Wd, 7), rtol=1e-5, atol=2e-4)


def attn_ref(q, k, v, views, heads):
    """fp64 softmax attention per view/head from the 16-bit operands; q [Rq, H*64], k/v [Rk, H*64]."""
    out = torch.zeros(q.shape, dtype=torch.float64)
    for (q0, nq, k0, nk, slo, shi) in views:
        keep = torch.ones(nk, dtype=torch.bool)
        keep[slo:shi] = False
        for h in range(heads):
            qq = q[q0:q0 + nq, h * 64:(h + 1) * 64].double()
            kk = k[k0:k0 + nk, h * 64:(h + 1) * 64].double()[keep]
            vv = v[k0:k0 + nk, h * 64:(h + 1) * 64].double()[keep]
            p = torch.softmax(qq @ kk.t() / 8.0, dim=-1)
            out[q0:q0 + nq, h * 64:(h + 1) * 64] = p @ vv
    return out


ATT_CASES = {
    # name: (heads, views [(q0,nq,k0,nk,skip_lo,skip_hi)], total q rows, total k rows, self?)
    "sa_196x2": (2, [(0, 196, 0, 196, 0, 0), (196, 196, 196, 196, 0, 0)], 392, 392, True),
    "sa_768": (3, [(0, 768, 0, 768, 0, 0)], 768, 768, True),
    "sa_tiny12": (2, [(0, 12, 0, 12, 0, 0), (12, 12, 12, 12, 0, 0), (24, 12, 24, 12, 0, 0)], 36, 36, True),
    "ca_tail_skip": (2, [(0, 100, 0, 1000, 300, 496), (100, 100, 0, 1000, 496, 692)], 200, 1000, False),
    "ca_aligned_skip": (2, [(0, 130, 0, 640, 128, 320)], 130, 640, False),
    "ca_skip_to_end": (1, [(0, 64, 0, 500, 304, 500)], 64, 500, False),
    "ca_skip_from_start": (1, [(0, 70, 0, 392, 0, 196), (70, 70, 0, 392, 196, 392)], 140, 392, False),
    "ca_long": (12, [(0, 768, 0, 4000, 0, 0)], 768, 4000, False),
}


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", list(ATT_CASES))
def test_attention(lib, dt, case):
    heads, views, Rq, Rk, is_self = ATT_CASES[case]
    D = heads * 64
    g = torch.Generator(device="cuda").manual_seed(11)
    tdt, u = DT[dt][1], DT[dt][2]
    if is_self:  # packed qkv rows like the fused projection writes them: [R, 3D]
        qkv = (torch.randn((Rq, 3 * D), device="cuda", generator=g) * 1.5).to(tdt)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    else:        # q [Rq, D], memory rows [Rk, 2D] = K|V
        q = (torch.randn((Rq, D), device="cuda", generator=g) * 1.5).to(tdt)
        kvm = (torch.randn((Rk, 2 * D), device="cuda", generator=g) * 1.5).to(tdt)
        k, v = kvm[:, :D], kvm[:, D:]
    o = torch.full((Rq, D), float("nan"), device="cuda", dtype=tdt)
    tab = torch.tensor(views, dtype=torch.int32, device="cuda")
    lib.check(lib.load().must3r_hip_op_attention(DT[dt][0], P(q), P(k), P(v), P(o), q.stride(0), k.stride(0), v.stride(0),
                                                 o.stride(0), heads, P(tab), len(views), max(vw[1] for vw in views), 0, None, 0,
                                                 stream()))
    torch.cuda.synchronize()
    ref = attn_ref(q.cpu(), k.cpu(), v.cpu(), views, heads)
    e = rel_inf(o.cpu(), ref)
    assert torch.isfinite(o.float()).all()
    assert e < 8 * u, e  # P and O are rounded to 16 bit; everything else is fp32
    # split-KV (flash-decoding) variants of the same problem must agree with the single-pass result
    es = {}
    for ns in (2, 3, 5):
        o2 = torch.full((Rq, D), float("nan"), device="cuda", dtype=tdt)
        nb = lib.load().must3r_hip_attention_scratch_bytes(ns, Rq, heads)
        ws = torch.empty((nb,), dtype=torch.uint8, device="cuda")
        lib.check(lib.load().must3r_hip_op_attention(DT[dt][0], P(q), P(k), P(v), P(o2), q.stride(0), k.stride(0), v.stride(0),
                                                     o2.stride(0), heads, P(tab), len(views), max(vw[1] for vw in views), ns,
                                                     P(ws), Rq, stream()))
        torch.cuda.synchronize()
        es[ns] = rel_inf(o2.cpu(), ref)
        assert torch.isfinite(o2.float()).all() and es[ns] < 8 * u, (ns, es[ns])
    record("attention", dt=dt, case=case, err=e, split_err=es)


@pytest.mark.parametrize("case", ["sa_196x2", "sa_768", "sa_tiny12", "ca_tail_skip", "ca_aligned_skip", "ca_skip_from_start", "ca_long"])
def test_attention_fp8(lib, case):
    """fp8 attention operands (BASELINE.json configs[4]): Q and K are e4m3 bytes and meet in v_mfma_scale_f32_32x32x64_f8f6f4;
    V, the softmax numerators and O stay fp16.  Reference: fp64 attention on the SAME (dequantised) operands, so what is measured
    is the kernel (layouts, masks, split-KV) and the fp16 rounding of P -- the operand quantisation itself is the model tests'."""
    from must3r_amd import _lib as _l
    if not _l.has_fp8_attention():
        pytest.skip("the e4m3 attention path is parked: built only with make EXTRA=-DM3R_ATTN_FP8 (include/must3r_hip.h)")
    heads, views, Rq, Rk, is_self = ATT_CASES[case]
    D = heads * 64
    g = torch.Generator(device="cuda").manual_seed(17)
    f8 = torch.float8_e4m3fn
    if is_self:
        qk = (torch.randn((Rq, 2 * D), device="cuda", generator=g) * 1.5).to(f8)
        q, k = qk[:, :D], qk[:, D:]
        v = (torch.randn((Rq, 3 * D), device="cuda", generator=g) * 1.5).half()[:, 2 * D:]
    else:   # memory rows [K e4m3 (D bytes) | V fp16 (2 D bytes)]
        q = (torch.randn((Rq, D), device="cuda", generator=g) * 1.5).to(f8)
        rows = torch.empty((Rk, 3 * D), dtype=torch.uint8, device="cuda")
        k = rows[:, :D].view(f8)
        v = rows[:, D:].view(torch.float16)
        k.copy_((torch.randn((Rk, D), device="cuda", generator=g) * 1.5).to(f8))
        v.copy_((torch.randn((Rk, D), device="cuda", generator=g) * 1.5).half())
    assert k.stride(0) == (2 * D if is_self else 3 * D) and v.stride(0) == (3 * D if is_self else 3 * D // 2)
    ref = attn_ref(q.float().cpu(), k.float().cpu(), v.float().cpu(), views, heads)
    tab = torch.tensor(views, dtype=torch.int32, device="cuda")
    errs = {}
    for ns in (0, 3):
        o = torch.full((Rq, D), float("nan"), device="cuda", dtype=torch.float16)
        ws = None
        if ns:
            ws = torch.empty((lib.load().must3r_hip_attention_scratch_bytes(ns, Rq, heads),), dtype=torch.uint8, device="cuda")
        lib.check(lib.load().must3r_hip_op_attention(lib.F16 | lib.ATTN_FP8, P(q), P(k), P(v), P(o), q.stride(0), k.stride(0),
                                                     v.stride(0), o.stride(0), heads, P(tab), len(views), max(vw[1] for vw in views),
                                                     ns, P(ws) if ns else None, Rq if ns else 0, stream()))
        torch.cuda.synchronize()
        assert torch.isfinite(o.float()).all()
        errs[ns] = rel_inf(o.cpu(), ref)
    record("attention_fp8", case=case, err=errs[0], split_err=errs[3])
    assert max(errs.values()) < 8 * 2.0 ** -11, errs      # exact e4m3 products, fp32 sums: what is left is the fp16 rounding of P and O


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_attention_running_max_jump(lib, dt):
    """Force the online-softmax rescale branch: one key late in the sequence dominates one query row."""
    heads, nq, nk = 1, 64, 512
    g = torch.Generator(device="cuda").manual_seed(13)
    tdt, u = DT[dt][1], DT[dt][2]
    q = torch.randn((nq, 64), device="cuda", generator=g).to(tdt)
    kv = torch.randn((nk, 128), device="cuda", generator=g).to(tdt)
    kv[400, :64] = (q[5].float() * 6).to(tdt)   # q5 . k400 >> every other score, in the 7th tile
    kv[3, :64] = (q[40].float() * 6).to(tdt)    # and one in the first tile
    views = [(0, nq, 0, nk, 0, 0)]
    o = torch.empty((nq, 64), device="cuda", dtype=tdt)
    tab = torch.tensor(views, dtype=torch.int32, device="cuda")
    lib.check(lib.load().must3r_hip_op_attention(DT[dt][0], P(q), P(kv), P(kv[:, 64:]), P(o), 64, 128, 128, 64, heads, P(tab), 1, nq,
                                                 0, None, 0, stream()))
    torch.cuda.synchronize()
    ref = attn_ref(q.cpu(), kv[:, :64].cpu(), kv[:, 64:].cpu(), views, heads)
    e = rel_inf(o.cpu(), ref)
    record("attention_jump", dt=dt, err=e)
    assert e < 8 * u, e


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("skip", [(0, 0), (300, 720)])   # an excluded key range (the MUSt3R own-token rule): partial tiles at both ends, the +17 spike (key 700) and the +11.5 one (500) inside it
@pytest.mark.parametrize("geom", [(1, 1), (12, 16)])   # (heads, views): one group = the 16-rows-per-wave form, 192 groups = the 32-row form (attention_is_small)
def test_attention_reference_moves_on_tile_row_sums(lib, dt, geom, skip):
    """r05 rule of attn3_kernel: the softmax references move when the row sums of a 64-key tile leave [0, 4096] (or are not finite), not on score maxima.
    Exact scores (q = unit vectors, four key columns carry the score): a staircase that climbs 5 log2 units per tile (the reference lags and catches up
    every other tile), a spike that overflows the 16-bit P (+17), one that overflows the fp32 exp2 (+140), one that stays below the trigger (+11.5:
    P = 2896 accumulates against the old reference) followed by scores 30 below it -- the four patterns interleaved inside every wave."""
    heads, nviews = geom
    nq, nk = 64, (1024 if skip == (0, 0) else 1000)   # with the excluded range also a partial last tile
    tdt, u = DT[dt][1], DT[dt][2]
    g = torch.Generator(device="cuda").manual_seed(29)
    per = 8.0 / 1.4426950408889634            # key column value of ONE log2 unit of score (softmax scale 1 / 8)
    q1 = torch.zeros((nq, 64), device="cuda")
    q1[torch.arange(nq), torch.arange(nq) % 4] = 1.0
    k1 = torch.zeros((nk, 64), device="cuda")
    k1[:, 0] = (torch.arange(nk, device="cuda") // 64).float() * 5.0 * per
    k1[700, 1] = 17.0 * per
    k1[900, 2] = 140.0 * per
    k1[500, 3] = 11.5 * per
    k1[576:, 3] = -30.0 * per
    q = q1.repeat(nviews, heads).to(tdt)
    k = k1.repeat(1, heads).to(tdt)
    v = torch.randn((nk, heads * 64), device="cuda", generator=g).to(tdt)
    D = heads * 64
    views = [(i * nq, nq, 0, nk, skip[0], skip[1]) for i in range(nviews)]
    o = torch.empty((nviews * nq, D), device="cuda", dtype=tdt)
    tab = torch.tensor(views, dtype=torch.int32, device="cuda")
    lib.check(lib.load().must3r_hip_op_attention(DT[dt][0], P(q), P(k), P(v), P(o), D, D, D, D, heads, P(tab), nviews, nq, 0, None, 0, stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    ref = attn_ref(q.cpu(), k.cpu(), v.cpu(), views, heads)
    per_pattern = [rel_inf(o.cpu()[i::4], ref[i::4]) for i in range(4)]
    record("attention_row_sum_rule", dt=dt, geom=list(geom), skip=list(skip), per_pattern=per_pattern)
    assert max(per_pattern) < 8 * u, per_pattern


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("MC", [(7, 128), (1000, 768), (513, 1024)])
def test_layernorm(lib, dt, MC):
    M, Cc = MC
    g = torch.Generator(device="cuda").manual_seed(3)
    tdt, u = DT[dt][1], DT[dt][2]
    x = torch.randn((M, Cc), device="cuda", generator=g) * 3 + 0.5
    add = torch.randn((M, Cc), device="cuda", generator=g)
    w = torch.randn((Cc,), device="cuda", generator=g)
    b = torch.randn((Cc,), device="cuda", generator=g)
    o16 = torch.empty((M, Cc), device="cuda", dtype=tdt)
    olo = torch.empty((M, Cc), device="cuda", dtype=tdt)
    o32 = torch.empty((M, Cc), device="cuda")
    cp = torch.empty((M, Cc), device="cuda")
    L = lib.load()
    lib.check(L.must3r_hip_op_layernorm(DT[dt][0], P(x), P(add), P(w), P(b), P(o16), P(olo), P(o32), P(cp), M, Cc, 1e-6, stream()))
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm((x + add).double(), (Cc,), w.double(), b.double(), 1e-6)
    assert torch.equal(cp, x + add)
    e = rel_inf(o32, ref)
    record("layernorm", dt=dt, MC=MC, err=e)
    assert torch.allclose(o32.double(), ref, rtol=1e-5, atol=1e-5), e
    assert torch.equal(o16, o32.to(tdt))
    assert torch.allclose(o16.double() + olo.double(), o32.double(), rtol=4 * u * u, atol=1e-6)
    lib.check(L.must3r_hip_op_layernorm(DT[dt][0], P(x), None, P(w), P(b), P(o16), None, None, None, M, Cc, 1e-5, stream()))
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.double(), (Cc,), w.double(), b.double(), 1e-5)
    assert torch.allclose(o16.double(), ref, rtol=2 * u, atol=2 * u)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_im2col_cast_postprocess(lib, dt):
    from oracle import must3r_ref as R
    tdt = DT[dt][1]
    L = lib.load()
    img = torch.randn((2, 3, 48, 64), device="cuda")
    out = torch.empty((2 * 12, 768), device="cuda", dtype=tdt)
    lib.check(L.must3r_hip_op_im2col(DT[dt][0], P(img), P(out), 2, 48, 64, stream()))
    ref = torch.nn.functional.unfold(img, kernel_size=16, stride=16).transpose(1, 2).reshape(24, 768)
    torch.cuda.synchronize()
    assert torch.equal(out, ref.to(tdt))
    x = torch.randn((4096,), device="cuda") * 10
    hi = torch.empty_like(x, dtype=tdt)
    lo = torch.empty_like(x, dtype=tdt)
    lib.check(L.must3r_hip_op_cast(DT[dt][0], P(x), P(hi), P(lo), x.numel(), stream()))
    torch.cuda.synchronize()
    assert torch.equal(hi, x.to(tdt)) and torch.equal(lo, (x - x.to(tdt).float()).to(tdt))
    from must3r_amd.engine import postprocess
    pm = torch.randn((2, 16, 16, 7), device="cuda") * 2
    o = postprocess(pm)
    r = R.postprocess(pm.cpu())
    for k in r:
        assert torch.allclose(o[k].cpu(), r[k], rtol=1e-5, atol=1e-6), k


@pytest.mark.parametrize("M", [768, 196])
def test_gemm_ln_fold(lib, M):
    """LN fold of the one-view memory update (must3r_hip_op_gemm_lnfold): a residual GEMM leaves the new rows rounded to fp16 plus per
    row and 16-column fragment (sum, sum of squares); the Linear that follows multiplies the RAW rows by gamma (.) W and normalises after
    the product: epi(rstd (x W'^T - mu s) + c) = epi(LN(x) W^T + b).  Checked against fp64 for the three consumers of a decoder block
    (projq: store + scale, fc1: GELU, qkv: RoPE) and against the unfolded HIP route (LayerNorm kernel + GEMM)."""
    from oracle import must3r_ref as R
    D = 768
    g = torch.Generator(device="cuda").manual_seed(11 + M)
    L = lib.load()
    # ---- producer: x += a Wp^T + bp
    x0 = torch.randn((M, D), device="cuda", generator=g) * (1.0 + 3.0 * torch.rand((M, 1), device="cuda", generator=g)) + \
        0.7 * torch.randn((M, 1), device="cuda", generator=g)
    x0[:8] += 40.0          # rows whose mean is ~20 standard deviations: what the per-row shift (the previous mean) is for
    x0[8:16, :3] *= 300.0   # rows with a few massive channels
    a = torch.randn((M, D), device="cuda", generator=g).half()
    Wp = torch.randn((D, D), device="cuda", generator=g) / math.sqrt(D)
    bp = torch.randn((D,), device="cuda", generator=g)
    x = x0.clone()
    x16 = torch.empty((M, D), device="cuda", dtype=torch.float16)
    cp = torch.empty((M, D), device="cuda")
    st = torch.full((M, D // 16, 2), float("nan"), device="cuda")
    # without a shift (the first producer of a call): raw rows
    lib.check(L.must3r_hip_op_gemm_lnfold(1, lib.EPI_RESID_F32, P(a), P(_split_w(Wp)), P(bp), P(x), M, D, D, D, D, P(x16), P(cp), P(st),
                                          None, None, 0.0, None, 0, None, None, 0, 0, 0.0, 0, stream()))
    torch.cuda.synchronize()
    want = x0.double() + a.double() @ Wp.double().t() + bp.double()
    assert torch.allclose(x.double(), want, rtol=1e-5, atol=1e-4) and torch.equal(cp, x) and torch.equal(x16, x.half())
    # with the shift a previous consumer left (here: the row means of the old x): rows and sums are those of y = x - shift
    shift = x0.mean(1).contiguous()
    x = x0.clone()
    lib.check(L.must3r_hip_op_gemm_lnfold(1, lib.EPI_RESID_F32, P(a), P(_split_w(Wp)), P(bp), P(x), M, D, D, D, D, P(x16), P(cp), P(st),
                                          None, None, 0.0, P(shift), 0, None, None, 0, 0, 0.0, 0, stream()))
    torch.cuda.synchronize()
    assert torch.allclose(x.double(), want, rtol=1e-5, atol=1e-4) and torch.equal(cp, x)
    ysh = x - shift[:, None]
    assert torch.equal(x16, ysh.half())
    fr = ysh.double().view(M, D // 16, 16)
    assert torch.allclose(st[..., 0].double(), fr.sum(-1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(st[..., 1].double(), (fr * fr).sum(-1), rtol=1e-5, atol=1e-3)
    # ---- consumers
    gam = 1.0 + 0.3 * torch.randn((D,), device="cuda", generator=g)
    bet = 0.2 * torch.randn((D,), device="cuda", generator=g)
    ln = torch.nn.functional.layer_norm(x.double(), (D,), gam.double(), bet.double(), 1e-6)
    h16 = torch.empty((M, D), device="cuda", dtype=torch.float16)
    lib.check(L.must3r_hip_op_layernorm(1, P(x), None, P(gam), P(bet), P(h16), None, None, None, M, D, 1e-6, stream()))
    gh, gw = (24, 32) if M == 768 else (14, 14)
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    pos = torch.stack((ys.reshape(-1), xs.reshape(-1)), -1).contiguous().cuda()
    buf = (C.c_float * (64 * 32))()
    L.must3r_hip_rope_table(100.0, 1.0, 64, buf)
    tab = torch.tensor(list(buf), device="cuda")
    errs = {}
    for name, epi, N, scale in (("projq", lib.EPI_STORE16, 768, 0.18), ("fc1", lib.EPI_STORE16_GELU, 3072, 0.0), ("qkv", lib.EPI_QKV_ROPE, 2304, 0.18)):
        W = torch.randn((N, D), device="cuda", generator=g) / math.sqrt(D)
        b = torch.randn((N,), device="cuda", generator=g)
        Wg = W * gam
        s_n = Wg.double().sum(1).float()
        c_n = (W.double() @ bet.double() + b.double()).float()
        out = torch.empty((M, N), device="cuda", dtype=torch.float16)
        rope = epi == lib.EPI_QKV_ROPE
        sh = shift.clone() if name != "fc1" else torch.full_like(shift, float("nan"))
        lib.check(L.must3r_hip_op_gemm_lnfold(1, epi, P(x16), P(_split_w(Wg)), P(c_n), P(out), M, N, D, D, N, None, None, None, P(st), P(s_n),
                                              1e-6, P(sh), 1 if name == "fc1" else 0, P(pos) if rope else None, P(tab) if rope else None,
                                              2 * D if rope else 0, 64 if rope else 0, scale, D if scale else 0, stream()))
        plain = torch.empty((M, N), device="cuda", dtype=torch.float16)
        lib.check(L.must3r_hip_op_gemm(1, epi, P(h16), P(_split_w(W)), P(b), P(plain), M, N, D, D, N, P(pos) if rope else None,
                                       P(tab) if rope else None, 2 * D if rope else 0, 64 if rope else 0, None, 0, 0, 0, 0, 0, 0, 2, stream()))
        torch.cuda.synchronize()
        y = ln @ W.double().t() + b.double()
        if epi == lib.EPI_STORE16_GELU:
            y = torch.nn.functional.gelu(y)
        if rope:
            yy = y.cpu().view(1, M, 3, 12, 64)
            q, k, v = (yy[:, :, i].permute(0, 2, 1, 3).float() for i in range(3))
            pp = pos.cpu().view(1, M, 2)
            y = torch.stack((R.rope2d(q, pp), R.rope2d(k, pp), v), dim=2).permute(0, 3, 2, 1, 4).reshape(M, N).double().cuda()
        y_plain = y   # the unfolded route is called without the scale on the first D columns
        if scale:
            y = y.clone()
            y[:, :D] *= scale
        errs[name] = (rel_inf(out, y), rel_inf(plain, y_plain))
        # the consumer leaves the mean of the rows it normalised for the next producer (init: the buffer held nothing -> mean of y)
        want_sh = x.double().mean(1) if name != "fc1" else ysh.double().mean(1)
        assert torch.allclose(sh.double(), want_sh, rtol=1e-5, atol=1e-4), name
        assert errs[name][0] < max(1e-3, 2 * errs[name][1]), (name, errs[name])
    record("gemm_ln_fold", M=M, errs=errs)
