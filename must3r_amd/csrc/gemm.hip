// MFMA GEMM for gfx950:  out[M,N] = epi(A[M,K] . W[N,K]^T + bias)
//
// * 256 threads = 4 waves arranged 2 (m) x 2 (n); block tile BM x BN x 64, wave tile (BM/2) x (BN/2)
//   built from v_mfma_f32_16x16x32 fragments.  The WEIGHT tile is the MFMA A operand (i = n) and the
//   ACTIVATION tile the B operand (j = m), so a lane ends up with 4 consecutive output columns of one
//   row: 8-byte (16-bit out) / 16-byte (fp32 out) stores, float4 bias loads, and the 2-D RoPE pairs
//   (d, d+16) of a head live in the same lane (fragments nf, nf+1) -> RoPE is a pure-register epilogue.
// * HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip), double-buffered, one barrier per
//   K-tile; the next tile's DMA is in flight while the current one is multiplied.
// * LDS tiles are [rows][64] 16-bit (128-byte rows).  global_load_lds writes lane-linear, so the
//   bank-conflict swizzle (common.hpp swz) is applied to the per-lane SOURCE address and again on the
//   ds_read_b128 side (same involution).
// * blockIdx -> tile mapping is XCD-aware: each XCD (blockIdx % 8) walks a contiguous chunk of the tile grid in
//   grouped order (8 row-blocks x all column-blocks) so the panels its resident blocks share stay in its L2.
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"

namespace m3r {

// WS = 2: split-weight mode, W is [N, 2K] = [W_hi | W_lo]; every K-tile stages the activation tile once plus BOTH weight
// tiles, and each activation fragment feeds two MFMAs (acc += W_hi.a ; acc += W_lo.a).
template <class T, int BM, int BN, int WGM, int WGN, int EPI, int NST, int WS, int BK>
__global__ void __launch_bounds__(64 * WGM * WGN) gemm_kernel(const GemmArgs p) {
    typedef typename Vec<T>::v8 v8;
    typedef typename Vec<T>::v4 v4;
    static_assert(BK == 64 || BK == 32, "K-tile depth");
    constexpr int NW = WGM * WGN;               // waves per block
    constexpr int WM = BM / WGM, WN = BN / WGN;  // wave tile
    constexpr int CPR = BK / 8;                 // 16-byte chunks per tile row
    constexpr int RPP = 64 / CPR;               // rows moved by one wave-wide 1 KiB DMA instruction
    constexpr int PA = BM / (RPP * NW), PW = BN / (RPP * NW);  // DMA pieces per wave and tile
    static_assert(BM % (RPP * NW) == 0 && BN % (RPP * NW) == 0 && WM % 16 == 0 && (WN == 32 || WN == 64), "tile geometry");
    constexpr int MF = WM / 16, NF = WN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);   // [NST][BM][BK]
    T* sW = sA + NST * BM * BK;           // [NST][WS][BN][BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    // ---- XCD-aware bijective remap of the linear block id
    const int nbn = p.N / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    // grouped tile order inside the XCD's contiguous chunk: GM row-blocks x all column-blocks, row-block fastest.  The
    // ~64 tiles an XCD works on at once then touch ~8 activation panels and ~8 weight panels (both L2-resident) instead
    // of streaming the whole weight matrix once per row-block (PMC r01: fc1 fetched 452 MB for 40 MB of operands).
    constexpr int GM = 8;
    const int tpg = GM * nbn;
    const int gidx = bid / tpg;
    const int gfirst = gidx * GM;
    const int gsz = (nbm - gfirst < GM) ? nbm - gfirst : GM;
    const int gin = bid - gidx * tpg;
    const int m0 = (gfirst + gin % gsz) * BM;
    const int n0 = (gin / gsz) * BN;

    const int grp = blockIdx.y;   // grouped launch: independent problems of equal shape
    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + (size_t)grp * p.strideA;
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W) + (size_t)grp * p.strideW;
    const float* __restrict__ bias = p.bias ? p.bias + (size_t)grp * p.strideB : nullptr;
    void* const outp = p.out_table ? p.out_table[grp] : p.out;

    // ---- staging: one wave instruction moves RPP rows x (BK*2) bytes = 1 KiB
    const int nka = p.K / BK;
    const int srow = lane / CPR;
    const int pch = lane % CPR;
    const T* a_src[PA];
    const T* w_src[PW];
#pragma unroll
    for (int t = 0; t < PA; ++t) {
        const int r = (wave * PA + t) * RPP + srow;
        int gr = m0 + r;
        gr = gr < p.M ? gr : p.M - 1;
        a_src[t] = A + (size_t)gr * p.lda + swzk<BK>(r, pch) * 8;
    }
#pragma unroll
    for (int t = 0; t < PW; ++t) {
        const int r = (wave * PW + t) * RPP + srow;
        int gr = n0 + r;
        gr = gr < p.N ? gr : p.N - 1;
        w_src[t] = W + (size_t)gr * (size_t)(p.K * WS) + swzk<BK>(r, pch) * 8;
    }
    auto stage = [&](int kt, int buf) {
#pragma unroll
        for (int t = 0; t < PA; ++t)
            glds16(a_src[t] + kt * BK, sA + (buf * BM + (wave * PA + t) * RPP) * BK);
#pragma unroll
        for (int part = 0; part < WS; ++part)
#pragma unroll
            for (int t = 0; t < PW; ++t)
                glds16(w_src[t] + part * p.K + kt * BK, sW + ((buf * WS + part) * BN + (wave * PW + t) * RPP) * BK);
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15;   // fragment row supplied by this lane
    const int fg = lane >> 4;   // k-group (16-byte chunk) supplied by this lane
    const int nk = nka;

    auto compute = [&](int buf) {
        const T* a = sA + buf * BM * BK;
        const T* w = sW + buf * WS * BN * BK;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            v8 wf[WS][NF], af[MF];
            const int lc = ks * 4 + fg;
#pragma unroll
            for (int part = 0; part < WS; ++part)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const int r = wn * WN + j * 16 + fr;
                    wf[part][j] = *reinterpret_cast<const v8*>(w + (part * BN + r) * BK + swzk<BK>(r, lc) * 8);
                }
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int r = wm * WM + i * 16 + fr;
                af[i] = *reinterpret_cast<const v8*>(a + r * BK + swzk<BK>(r, lc) * 8);
            }
#pragma unroll
            for (int part = 0; part < WS; ++part)
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[i][j] = mfma16(wf[part][j], af[i], acc[i][j]);
        }
    };

    if constexpr (NST == 2) {
        // double buffer: the DMA of tile kt+1 is in flight while tile kt is multiplied
        stage(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's DMA for tile kt has landed
            __syncthreads();                     // ... everyone's has, and tile kt-1 is no longer being read
            if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
            compute(buf);
        }
    } else {
        // NST-deep ring with COUNTED waits: NST-1 tiles of DMA in flight per block.  Small-M GEMMs run ~1 block per CU,
        // so the per-tile L2/HBM round trip (not MFMA time) sets the pace unless several tiles overlap.
        // Each wave issues IPT global_load_lds per tile; before tile kt is read only the (NST-2) younger tiles may
        // still be outstanding -> s_waitcnt vmcnt((NST-2)*IPT), then a raw s_barrier (no vmcnt(0) drain).
        constexpr int IPT = PA + WS * PW;
        constexpr int PEND = (NST - 2) * IPT;
        static_assert(PEND < 64, "vmcnt field");
#pragma unroll
        for (int t = 0; t < NST - 1; ++t)
            if (t < nk) stage(t, t);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + NST - 2 < nk) __builtin_amdgcn_s_waitcnt(0x0f70 | (PEND & 15) | ((PEND >> 4) << 14));
            else __builtin_amdgcn_s_waitcnt(0x0f70);   // tail: fewer tiles behind this one, drain
            __builtin_amdgcn_s_barrier();
            const int nt = kt + NST - 1;
            if (nt < nk) {
                int nb = buf + NST - 1;
                nb = nb >= NST ? nb - NST : nb;
                stage(nt, nb);                         // overwrites the buffer read in iteration kt-1
            }
            compute(buf);
            buf = buf + 1 == NST ? 0 : buf + 1;
        }
    }

    // ---- epilogue: acc[i][j][r] = C[m = m0 + wm*WM + i*16 + fr][n = n0 + wn*WN + j*16 + fg*4 + r]
    const int nb = n0 + wn * WN + fg * 4;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = m0 + wm * WM + i * 16 + fr;
        if (m >= p.M) continue;
        f32x4 v[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            v[j] = acc[i][j];
            const bool nobias = (EPI == EPI_F32 || EPI == EPI_HEAD) && p.accumulate;
            if (bias != nullptr && !nobias) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias + nb + j * 16);
                v[j] += b;
            }
        }
        if constexpr (EPI == EPI_QKV_ROPE) {
            // wave tile is 32- or 64-column aligned inside a 64-wide head: fragments (2q, 2q+1) are the
            // rotate-half pair of one 32-wide half; even halves rotate by y, odd halves by x.
            if (n0 + wn * WN < p.rope_cols) {
                const long long py = p.pos[(size_t)m * 2 + 0];
                const long long px = p.pos[(size_t)m * 2 + 1];
#pragma unroll
                for (int q = 0; q < NF / 2; ++q) {
                    const int nh = n0 + wn * WN + q * 32;
                    int pp = (int)(((nh >> 5) & 1) ? px : py);
                    pp = pp < 0 ? 0 : (pp >= p.rope_npos ? p.rope_npos - 1 : pp);
                    const float* tb = p.rope_tab + ((size_t)pp * 16 + fg * 4) * 2;
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(tb);      // cos0 sin0 cos1 sin1
                    const f32x4 t1 = *reinterpret_cast<const f32x4*>(tb + 4);  // cos2 sin2 cos3 sin3
                    const float cs[4] = {t0[0], t0[2], t1[0], t1[2]};
                    const float sn[4] = {t0[1], t0[3], t1[1], t1[3]};
                    const f32x4 x0 = v[2 * q], x1 = v[2 * q + 1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[2 * q][r] = x0[r] * cs[r] - x1[r] * sn[r];
                        v[2 * q + 1][r] = x1[r] * cs[r] + x0[r] * sn[r];
                    }
                }
            }
        }
        if constexpr (EPI == EPI_STORE16 || EPI == EPI_QKV_ROPE) {
            if (p.out_scale != 0.f && n0 + wn * WN < p.scale_cols) {   // scale_cols is a multiple of 64: wave-uniform
#pragma unroll
                for (int j = 0; j < NF; ++j) v[j] *= p.out_scale;
            }
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int n = nb + j * 16;
            if constexpr (EPI == EPI_STORE16 || EPI == EPI_QKV_ROPE) {
                *reinterpret_cast<v4*>(reinterpret_cast<T*>(outp) + (size_t)m * p.ldc + n) = cvt4<T>(v[j]);
            } else if constexpr (EPI == EPI_STORE16_GELU) {
                f32x4 g;
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = gelu_erf(v[j][r]);
                *reinterpret_cast<v4*>(reinterpret_cast<T*>(outp) + (size_t)m * p.ldc + n) = cvt4<T>(g);
            } else if constexpr (EPI == EPI_RESID_F32) {
                f32x4* o = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outp) + (size_t)m * p.ldc + n);
                *o = *o + v[j];
            } else if constexpr (EPI == EPI_F32) {
                f32x4* o = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outp) + (size_t)m * p.ldc + n);
                f32x4 x = v[j];
                if (p.accumulate) {
                    x += *o;
                } else if (p.bias2 != nullptr && m >= p.row_start2) {
                    x += *reinterpret_cast<const f32x4*>(p.bias2 + n);
                }
                *o = x;
            } else if constexpr (EPI == EPI_HEAD) {
                // permuted feature n = (i*16 + jj)*7 + c ; token t of view vv at grid (gy, gx)
                const int vv = m / p.ntok, t = m - vv * p.ntok;
                const int gy = t / p.gw, gx = t - gy * p.gw;
                const int pi = n / 112, rem = n - pi * 112;
                const size_t off = ((size_t)(vv * p.H + gy * 16 + pi) * p.Wimg + gx * 16) * 7 + rem;
                f32x4* o = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outp) + off);
                f32x4 x = v[j];
                if (p.accumulate) x += *o;
                *o = x;
            }
        }
    }
}

template <class T, int BM, int BN, int WGM, int WGN, int EPI, int NST, int WS, int BK = 64>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
    const int nbn = a.N / BN, nbm = (a.M + BM - 1) / BM;
    const size_t lds = (size_t)NST * (BM + WS * BN) * BK * sizeof(T);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, BM, BN, WGM, WGN, EPI, NST, WS, BK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WGM, WGN, EPI, NST, WS, BK>), dim3(nbm * nbn, a.batch > 1 ? a.batch : 1),
                       dim3(64 * WGM * WGN), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Tile selection (measured on MI355X, scripts/bench_gemm.py): two resident blocks per CU beat every larger tile that
// leaves one (128x128 3-stage, 256x128 with 4 or 8 waves: 465-613 TF/s vs 644 TF/s on the scene's big-batch shapes).
//   plain weights : 128x128x64, 2 stages (64 KB)  for chip-filling grids, 64x64x64 4-stage ring (64 KB) otherwise
//   split weights : 128x64 (+64 lo) 2 stages (64 KB) / 64x64 (+64 lo) 3-stage ring (72 KB)
// K-tile depth 32 (the BK template parameter; 3-5 resident blocks per CU) was also measured: 605-626 TF/s plain,
// 397-419 vs 409 TF/s split -- no gain, so only BK = 64 is instantiated.
// minimum number of big tiles for the big-tile kernel (tunable for experiments: M3R_GEMM_MIN_BIG / _MIN_BIG_SPLIT)
static long min_big(bool split) {
    static long v[2] = {-1, -1};
    if (v[split] < 0) {
        const char* e = getenv(split ? "M3R_GEMM_MIN_BIG_SPLIT" : "M3R_GEMM_MIN_BIG");
        v[split] = e ? atol(e) : (split ? 384 : 192);
    }
    return v[split];
}

template <class T, int EPI>
static int launch_epi(const GemmArgs& a, hipStream_t s, const char** err) {
    const long nb = a.batch > 1 ? a.batch : 1;
    int rc;
    if (a.wsplit == 2) {
        if constexpr (sizeof(T) == 2 && std::is_same<T, f16_t>::value) {
            const long tiles = (long)((a.M + 127) / 128) * (a.N / 64) * nb;
            if (tiles >= min_big(true)) rc = launch_cfg<T, 128, 64, 2, 2, EPI, 2, 2>(a, s);
            else rc = launch_cfg<T, 64, 64, 2, 2, EPI, 3, 2>(a, s);
        } else {
            *err = "gemm: split weights are only built for fp16";
            return 1;
        }
    } else {
        const bool n128 = (a.N % 128) == 0;
        const long tiles128 = (long)((a.M + 127) / 128) * (a.N / 128) * nb;
        if (n128 && tiles128 >= min_big(false)) rc = launch_cfg<T, 128, 128, 2, 2, EPI, 2, 1>(a, s);
        else rc = launch_cfg<T, 64, 64, 2, 2, EPI, 4, 1>(a, s);
    }
    if (rc) *err = "gemm: kernel launch failed";
    return rc;
}

template <class T>
static int launch_t(Epi epi, const GemmArgs& a, hipStream_t s, const char** err) {
    switch (epi) {
        case EPI_STORE16: return launch_epi<T, EPI_STORE16>(a, s, err);
        case EPI_STORE16_GELU: return launch_epi<T, EPI_STORE16_GELU>(a, s, err);
        case EPI_QKV_ROPE: return launch_epi<T, EPI_QKV_ROPE>(a, s, err);
        case EPI_RESID_F32: return launch_epi<T, EPI_RESID_F32>(a, s, err);
        case EPI_F32: return launch_epi<T, EPI_F32>(a, s, err);
        case EPI_HEAD: return launch_epi<T, EPI_HEAD>(a, s, err);
        default: *err = "gemm: bad epilogue"; return 1;
    }
}

int launch_gemm(DType dt, Epi epi, const GemmArgs& a, hipStream_t s, const char** err) {
    if (a.M <= 0) return 0;
    if (a.N % 64 != 0 || a.K % 64 != 0 || a.N <= 0 || a.K <= 0) { *err = "gemm: N and K must be multiples of 64"; return 1; }
    if (a.lda % 8 != 0 || (epi != EPI_HEAD && a.ldc % 4 != 0)) { *err = "gemm: lda%8 / ldc%4 alignment"; return 1; }
    if (epi == EPI_QKV_ROPE && (a.pos == nullptr || a.rope_tab == nullptr || a.rope_cols % 64 != 0)) {
        *err = "gemm: rope epilogue needs pos, table and 64-aligned rope_cols"; return 1;
    }
    if (epi == EPI_HEAD && (a.N % 112 != 0 || a.ntok <= 0 || a.gw <= 0)) { *err = "gemm: head epilogue geometry"; return 1; }
    return dt == DT_BF16 ? launch_t<bf16_t>(epi, a, s, err) : launch_t<f16_t>(epi, a, s, err);
}

}  // namespace m3r
