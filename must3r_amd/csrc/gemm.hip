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
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"
#include "options.hpp"

namespace m3r {

#ifdef GEMM_TRACE
__device__ unsigned long long g_gemm_trace[64 * 8];   // scripts/probes/gemm_trace.hip: cycle stamps of block 0 / thread 0
__device__ unsigned long long g_gemm256_trace[2][64 * 8];   // scripts/probes/gemm256_trace.hip: block 0, lane 0 of wave 0 (group 0) / wave 4 (group 1)
#define M3R_STAMP256(slot) do { if (blockIdx.x == 0 && lane == 0 && wc == 0 && t < 64) g_gemm256_trace[wr][t * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
// gemm256p_kernel (scripts/probes/gemm256p_trace.hip): 16 stamps per K-tile, K-tiles 0..31, s_memtime (shader cycles)
#define M3R_STAMPP(slot) do { if (blockIdx.x == p.trace_block && lane == 0 && wc == 0 && t < 32) g_gemm256_trace[wr][t * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define M3R_STAMP256(slot) do { } while (0)
#define M3R_STAMPP(slot) do { } while (0)
#endif

// One output row segment of a lane: v[j][r] = C[m][n = nw0 + j*16 + fg*4 + r] before bias; nw0 = first column of the wave tile.
// Shared by every tile geometry so that all of them round identically (built with -ffp-contract=off).
// The small-M kernels load the bias fragments (and, for the residual epilogue, the old x values) BEFORE their K loop: after the loop
// they would be one or two dependent L2 round trips (~0.5-1 us) at the end of a 7-14 us launch.
// out / out_table[g] are global memory; telling the compiler so keeps the epilogue's accesses global_* instead of flat_* (a pointer read from
// out_table has no address space the compiler can see)
template <class V> __device__ __forceinline__ void st_global(void* ptr, const V v) {
    *reinterpret_cast<__attribute__((address_space(1))) V*>((unsigned long long)ptr) = v;
}
template <class V> __device__ __forceinline__ V ld_global(const void* ptr) {
    return *reinterpret_cast<const __attribute__((address_space(1))) V*>((unsigned long long)ptr);
}
template <int NF> struct EpiPre {
    f32x4 b[NF];   // bias columns of this lane (zero when the epilogue adds none)
    f32x4 x[NF];   // old residual values (EPI_RESID_F32 only)
};
template <int NF> struct RopePre {
    f32x4 t0[NF / 2 > 0 ? NF / 2 : 1], t1[NF / 2 > 0 ? NF / 2 : 1];   // the (cos, sin) table entries of the row's position, per 32-column half
};
template <class T, int EPI, int NF>
__device__ __forceinline__ void epilogue_prefetch(const GemmArgs& p, void* const outp, const float* __restrict__ bias, const int m,
                                                  const int nw0, const int fg, EpiPre<NF>& pre) {
    const int nb = nw0 + fg * 4;
    const bool nobias = (EPI == EPI_F32 || EPI == EPI_HEAD) && p.accumulate;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        pre.b[j] = (bias != nullptr && !nobias) ? *reinterpret_cast<const f32x4*>(bias + nb + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == EPI_RESID_F32) {
            const int mm = m < p.M ? m : p.M - 1;
            pre.x[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(outp) + (size_t)mm * p.ldc + nb + j * 16);
        }
    }
}

// LN fold, producer side: the new residual values of one lane (4 columns of row m) -> 16-bit copy, optional fp32 copy, and the
// (sum, sum of squares) of the row's 16-column fragment (the four lanes fg = 0..3 of a row hold one fragment)
template <class T>
__device__ __forceinline__ void ln_fold_emit(const GemmArgs& p, const int m, const int n, const int fg, const f32x4 xn, const float shift) {
    typedef typename Vec<T>::v4 v4;
    if (p.copy32_out) *reinterpret_cast<f32x4*>(p.copy32_out + (size_t)m * p.ldc + n) = xn;
    const f32x4 y = xn - shift;   // shift = the row mean the previous consumer measured (0 without one)
    if (p.x16_out) *reinterpret_cast<v4*>(reinterpret_cast<T*>(p.x16_out) + (size_t)m * p.ldc + n) = cvt4_sat<T>(y);
    if (p.stats_out) {
        const float s1 = quad_row_sum((y[0] + y[1]) + (y[2] + y[3]));
        const float s2 = quad_row_sum((y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]));
        if (fg == 0) {
            float* d = p.stats_out + ((size_t)m * (p.N >> 4) + (n >> 4)) * 2;
            d[0] = s1;
            d[1] = s2;
        }
    }
}

// LN fold, consumer side: (mu, rstd) of the block's BM rows from the producer's fragment sums, left in LDS at sm[BM*TPR*2 + 2*row].
// NT threads, TPR = NT / BM threads per row, each adds its share of the K/16 fragments in order, thread 0 of a row adds the TPR
// partial sums in order: deterministic.  Raw barriers (LDS-DMA of the ring prologue may be in flight; __syncthreads would drain it).
constexpr int LNF_SLOTS = 48;   // K = 768 (the decoder width): 48 fragments per row; launch_gemm refuses anything else
template <int BM, int NT>
__device__ __forceinline__ void ln_fold_rows(const GemmArgs& p, const int m0, float* sm, const int tid, const bool first_col) {
    constexpr int TPR = NT / BM;
    constexpr int PER = LNF_SLOTS / TPR;   // fragments per thread: 12 / 8 / 6 / 4 (all loads issued before the first add)
    static_assert(TPR * BM == NT && PER * TPR == LNF_SLOTS && PER % 2 == 0, "threads per row");
    const int row = tid / TPR, part = tid - row * TPR;
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    const float* src = p.ln_stats + ((size_t)m * LNF_SLOTS + (size_t)part * PER) * 2;
    f32x4 q[PER / 2];
#pragma unroll
    for (int i = 0; i < PER / 2; ++i) q[i] = *reinterpret_cast<const f32x4*>(src + i * 4);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < PER / 2; ++i) {
        s1 += q[i][0]; s2 += q[i][1];
        s1 += q[i][2]; s2 += q[i][3];
    }
    sm[(row * TPR + part) * 2] = s1;
    sm[(row * TPR + part) * 2 + 1] = s2;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (part == 0) {
        float a = 0.f, b = 0.f;
        for (int t = 0; t < TPR; ++t) {
            a += sm[(row * TPR + t) * 2];
            b += sm[(row * TPR + t) * 2 + 1];
        }
        const float inv = 1.0f / (float)p.K;
        const float mu = a * inv;
        float var = b * inv - mu * mu;
        var = var > 0.f ? var : 0.f;
        sm[BM * TPR * 2 + row * 2] = mu;
        sm[BM * TPR * 2 + row * 2 + 1] = rsqrtf(var + p.ln_eps);
        // the blocks of column 0 leave the current row mean for the next producer (the rows were shifted by the previous estimate)
        if (first_col && p.ln_shift != nullptr && m0 + row < p.M)
            p.ln_shift[m0 + row] = (p.ln_shift_init ? 0.f : p.ln_shift[m0 + row]) + mu;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
template <int BM, int NT> constexpr size_t ln_fold_lds_bytes() { return (size_t)(BM * (NT / BM) * 2 + BM * 2) * sizeof(float); }

// ---- r06: the LN fold on the chip-filling 256 x 256 tiles (GemmArgs::fold256; gemm256s_kernel / gemm256p_kernel<.., FOLD = 1>) ----
// Statistics travel per row and 64-column WAVE TILE ([M][N/64][2]: a quarter of the legacy per-fragment layout's bytes): 12 (K = 768) or 16 (K = 1024) slots per row.
// Consumer, in front of its first DMA: two threads per row request their halves of the row's slots (16-byte loads) + the row's current shift (+ the caller's s_n
// columns).  INLINE ASM: a load the compiler knows about makes it wait vmcnt(0) in front of the first use -- i.e. for the whole prologue's DMA, K-tile 1 included
// (measured: +2.5 us per tile).  The registers are "defined" for the compiler at the asm statement and really filled when the loads land: they must not be read,
// copied or spilled before fold256_landed() -- placed behind the prologue's counted wait, which these older loads are covered by (in-order vmcnt) -- has
// re-defined them in place (the pattern of gemm256s_kernel's load_ix / take_ix; checked in the listing: scripts/checks/fold256_regs.py).
struct Fold256In { f32x4 q0, q1, q2, q3, s; float sh; };
__device__ __forceinline__ void fold256_issue(const GemmArgs& p, const int m0, const int n0, const int tid, Fold256In& f) {
    const int nsl = p.K >> 6, half = nsl >> 1;   // launch_gemm: 12 or 16 slots per row
    int m = m0 + (tid >> 1);
    m = m < p.M ? m : p.M - 1;
    const float* src = p.ln_stats + ((size_t)m * nsl + (size_t)(tid & 1) * half) * 2;
    const float* src3 = src + (half == 8 ? 12 : 8);   // K = 768: three loads cover the half row; the fourth repeats the third and is dropped
    const float* shp = p.ln_shift + m;
    const float* sp = p.ln_s + n0 + (tid & 63) * 4;
    asm volatile("global_load_dwordx4 %0, %6, off\n\tglobal_load_dwordx4 %1, %6, off offset:16\n\tglobal_load_dwordx4 %2, %6, off offset:32\n\t"
                 "global_load_dwordx4 %3, %7, off\n\tglobal_load_dword %4, %8, off\n\tglobal_load_dwordx4 %5, %9, off"
                 : "=&v"(f.q0), "=&v"(f.q1), "=&v"(f.q2), "=&v"(f.q3), "=&v"(f.sh), "=&v"(f.s)
                 : "v"(src), "v"(src3), "v"(shp), "v"(sp)
                 : "memory");
}
__device__ __forceinline__ void fold256_landed(Fold256In& f) {
    asm volatile("" : "+v"(f.q0), "+v"(f.q1), "+v"(f.q2), "+v"(f.q3), "+v"(f.sh), "+v"(f.s));
}
// ... and, with the rest of the prologue's DMA in flight, add them up in slot order (the partner lane's half through a DPP quad permute: both lanes hold the row's
// (mean, 1/sigma) of the SHIFTED row -- all the fold needs, see GemmArgs::ln_shift); the column-0 blocks leave shift + mean = the row's current mean.
__device__ __forceinline__ void fold256_finish(const GemmArgs& p, const int m0, const int n0, const int tid, const Fold256In& f, float& mu, float& rstd) {
    float s1 = 0.f, s2 = 0.f;
    s1 += f.q0[0]; s2 += f.q0[1]; s1 += f.q0[2]; s2 += f.q0[3];
    s1 += f.q1[0]; s2 += f.q1[1]; s1 += f.q1[2]; s2 += f.q1[3];
    s1 += f.q2[0]; s2 += f.q2[1]; s1 += f.q2[2]; s2 += f.q2[3];
    if ((p.K >> 7) == 8) {   // (kernel-uniform)
        s1 += f.q3[0]; s2 += f.q3[1]; s1 += f.q3[2]; s2 += f.q3[3];
    }
    const float o1 = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s1), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    const float o2 = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s2), 0xB1, 0xf, 0xf, false));
    const float a = (tid & 1) ? o1 + s1 : s1 + o1;   // first half + second half, in both lanes
    const float b = (tid & 1) ? o2 + s2 : s2 + o2;
    const float inv = 1.0f / (float)p.K;
    mu = a * inv;
    float var = b * inv - mu * mu;
    var = var > 0.f ? var : 0.f;
    rstd = rsqrtf(var + p.ln_eps);
    if (n0 == 0 && !(tid & 1) && m0 + (tid >> 1) < p.M) st_global<float>(p.ln_shift + m0 + (tid >> 1), f.sh + mu);
}
// Producer: one row fragment of a 64-column wave tile (NF = 4), new fp32 values xn of this lane -> the shifted row's 16-bit copy (whole 128-byte lines, the
// lane exchanges of the 16-bit-store epilogues) and the wave tile's (sum, sum of squares) of that row.  Needs all 16 rows of the fragment (M % 256 == 0).
template <class T>
__device__ __forceinline__ void fold256_emit(const GemmArgs& p, const int m, const int nw0, const int fg, const f32x4 (&xn)[4], const float shift) {
    typedef typename Vec<T>::v4 v4;
    f32x4 y[4];
    float s1 = 0.f, s2 = 0.f, t1 = 0.f, t2 = 0.f;   // (two chains each; squares by fused multiply-add: the file is built with -ffp-contract=off)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        y[j] = xn[j] - shift;
        s1 += y[j][0] + y[j][1];
        t1 += y[j][2] + y[j][3];
        s2 = __builtin_fmaf(y[j][0], y[j][0], s2); t2 = __builtin_fmaf(y[j][1], y[j][1], t2);
        s2 = __builtin_fmaf(y[j][2], y[j][2], s2); t2 = __builtin_fmaf(y[j][3], y[j][3], t2);
    }
    s1 = quad_row_sum(s1 + t1);
    s2 = quad_row_sum(s2 + t2);
    if (fg == 0) st_global<f32x2>(p.stats_out + ((size_t)m * (p.N >> 6) + (nw0 >> 6)) * 2, f32x2{s1, s2});
    const int lane = (int)__lane_id();
    u32x4 o[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const v4 h0 = cvt4_sat<T>(y[2 * q]), h1 = cvt4_sat<T>(y[2 * q + 1]);
        unsigned a[2], c[2];
        __builtin_memcpy(a, &h0, 8);
        __builtin_memcpy(c, &h1, 8);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            auto w1 = __builtin_amdgcn_permlane32_swap(a[d], c[d], false, false);
            auto w2 = __builtin_amdgcn_permlane16_swap(w1[0], w1[1], false, false);
            o[q][d] = w2[0];
            o[q][2 + d] = w2[1];
        }
    }
    u32x4 x, z;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        x[d] = (unsigned)__builtin_amdgcn_update_dpp((int)o[0][d], (int)o[1][d], 0x128, 0xf, 0xc, false);
        z[d] = (unsigned)__builtin_amdgcn_update_dpp((int)o[1][d], (int)o[0][d], 0x128, 0xf, 0x3, false);
    }
    const int row0 = m - (lane & 15);
    const int nq = nw0 + ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8;
    T* const dst = reinterpret_cast<T*>(p.x16_out) + (size_t)(row0 + (lane & 7)) * p.ldc + nq + ((lane >> 3) & 1) * 32;
    st_global<u32x4>(dst, x);
    st_global<u32x4>(dst + (size_t)8 * p.ldc, z);
}

// Consumer, behind the K loop: acc <- acc / sigma_m - (mu_m / sigma_m) s_n, i.e. what epilogue_row's LN path computes per row ((acc - mu s) / sigma) in two
// operations per value instead of three; the epilogue then adds c_n as the bias.  The rows' (mean, 1/sigma) and the columns' s_n reach the lanes that hold the
// accumulators through LDS:
//   fold256_apply_staged (gemm256p_kernel: 32 KB of the CU's LDS are free beside its two 64 KB buffers): the kernel left them at `sm` in front of its K loop -- no
//     barrier, no global load here;
template <int NF, int MF>
__device__ __forceinline__ void fold256_apply_staged(const char* const sm, const int row_in_tile, const int col_in_tile, f32x4 (&acc)[MF][NF]) {
    const f32x2* const sm2 = reinterpret_cast<const f32x2*>(sm);
    const float* const ss = reinterpret_cast<const float*>(sm + 2048);
    f32x4 s4[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) s4[j] = *reinterpret_cast<const f32x4*>(ss + col_in_tile + j * 16);
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const f32x2 mr = sm2[row_in_tile + i * 16];
        const float nm = -mr[0] * mr[1];
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(s4[j][r], nm, acc[i][j][r] * mr[1]);
    }
}
//   fold256_apply (gemm256s_kernel: its two buffers are the whole LDS): s_n and the bias columns c_n are requested together (ONE round trip: the kernel's epilogue
//     would have paid it for the bias anyway; the caller hands `b` on as epilogue_tile's bpre), the statistics go through the first 2 KB of the idle buffers.
template <int NF, int MF>
__device__ __forceinline__ void fold256_apply(const GemmArgs& p, const float* __restrict__ bias, char* const smem, const int tid, const int row_in_tile, const int col,
                                              const float mu, const float rstd, f32x4 (&acc)[MF][NF], f32x4 (&b)[NF]) {
    f32x4 s4[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        s4[j] = ld_global<f32x4>(p.ln_s + col + j * 16);
        b[j] = ld_global<f32x4>(bias + col + j * 16);
    }
    f32x2* const sm2 = reinterpret_cast<f32x2*>(smem);
    if (!(tid & 1)) sm2[tid >> 1] = f32x2{mu, rstd};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const f32x2 mr = sm2[row_in_tile + i * 16];
        const float nm = -mr[0] * mr[1];
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(s4[j][r], nm, acc[i][j][r] * mr[1]);
    }
}

// LN = false: the LN-fold paths (consumer: ln_stats / ln_s; producer: x16_out / copy32_out / stats_out / ln_shift) are compiled out -- the
// chip-filling tile shapes never run them (launch_epi routes such calls to the small-tile kernels), and a load under a branch in front of
// every row fragment's stores makes the compiler wait for vmcnt(0) at the join, i.e. for the previous fragment's stores.
template <class T, int EPI, int NF, bool LN = true>
__device__ __forceinline__ void epilogue_row(const GemmArgs& p, void* const outp, const float* __restrict__ bias, const int m,
                                             const int nw0, const int fg, f32x4 (&v)[NF], const EpiPre<NF>* pre = nullptr,
                                             const float ln_mu = 0.f, const float ln_rstd = 0.f, const f32x4* b2pre = nullptr,
                                             const RopePre<NF>* rp = nullptr) {
    typedef typename Vec<T>::v4 v4;
    const int nb = nw0 + fg * 4;
    float ln_shift = 0.f;
    if constexpr (LN && (EPI == EPI_RESID_F32 || EPI == EPI_F32)) {
        if (p.ln_shift != nullptr && (p.x16_out != nullptr || p.stats_out != nullptr)) ln_shift = p.ln_shift[m];
    }
    if constexpr (LN && (EPI == EPI_STORE16 || EPI == EPI_STORE16_GELU || EPI == EPI_QKV_ROPE)) {
        if (p.ln_stats != nullptr) {   // LN fold: acc = sum_k x_k W'_nk  ->  rstd (acc - mu s_n); c_n comes in as the bias
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.ln_s + nb + j * 16);
                v[j] = (v[j] - s4 * ln_mu) * ln_rstd;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const bool nobias = (EPI == EPI_F32 || EPI == EPI_HEAD) && p.accumulate;
        if (pre != nullptr) {
            v[j] += pre->b[j];
        } else if (bias != nullptr && !nobias) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + nb + j * 16);
            v[j] += b;
        }
    }
    if constexpr (EPI == EPI_QKV_ROPE) {
        // wave tile is 32- or 64-column aligned inside a 64-wide head: fragments (2q, 2q+1) are the
        // rotate-half pair of one 32-wide half; even halves rotate by y, odd halves by x.
        if (nw0 < p.rope_cols) {
            const long long py = rp ? 0 : p.pos[(size_t)m * 2 + 0];
            const long long px = rp ? 0 : p.pos[(size_t)m * 2 + 1];
#pragma unroll
            for (int q = 0; q < NF / 2; ++q) {
                const int nh = nw0 + q * 32;
                int pp = (int)(((nh >> 5) & 1) ? px : py);
                pp = pp < 0 ? 0 : (pp >= p.rope_npos ? p.rope_npos - 1 : pp);
                const float* tb = p.rope_tab + ((size_t)pp * 16 + fg * 4) * 2;
                const f32x4 t0 = rp ? rp->t0[q] : *reinterpret_cast<const f32x4*>(tb);      // cos0 sin0 cos1 sin1
                const f32x4 t1 = rp ? rp->t1[q] : *reinterpret_cast<const f32x4*>(tb + 4);  // cos2 sin2 cos3 sin3
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
        if (p.out_scale != 0.f && nw0 < p.scale_cols) {   // scale_cols is a multiple of 64: wave-uniform
#pragma unroll
            for (int j = 0; j < NF; ++j) v[j] *= p.out_scale;
        }
    }
    if constexpr ((EPI == EPI_STORE16 || EPI == EPI_QKV_ROPE || EPI == EPI_STORE16_GELU) && NF % 2 == 0) {
        // 16-bit outputs: a lane's fragment is 4 columns = 8 bytes, the four lanes of a row cover 32 contiguous bytes per store -- a quarter
        // of a cache line, and the memory system takes such stores at ~2 TB/s (a 256 x 256 x 64 GEMM launch of 94 MB: 56 us against 11 us
        // without its stores, profiles/r03_gemm256k_fixed.txt).  Two cross-lane swaps per dword rotate (fragment parity j0, lane bit 5, lane bit 4)
        // so that a lane holds 8 consecutive columns of fragments (2q, 2q+1) and a row's four lanes 64 contiguous bytes: half as many stores,
        // 16 bytes per lane.  Values are untouched (the same bits land in the same places).
        const int lane = (int)__lane_id();
        u32x4 o[NF / 2];
#pragma unroll
        for (int q = 0; q < NF / 2; ++q) {
            f32x4 g0 = v[2 * q], g1 = v[2 * q + 1];
            if constexpr (EPI == EPI_STORE16_GELU) {
                g0 = gelu_erf4(g0);
                g1 = gelu_erf4(g1);
            }
            const v4 h0 = cvt4_sat<T>(g0), h1 = cvt4_sat<T>(g1);
            unsigned a[2], b[2];
            __builtin_memcpy(a, &h0, 8);
            __builtin_memcpy(b, &h1, 8);
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                auto s1 = __builtin_amdgcn_permlane32_swap(a[d], b[d], false, false);   // fragment parity <-> lane bit 5
                auto s2 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false); // (old lane bit 5) <-> lane bit 4
                o[q][d] = s2[0];
                o[q][2 + d] = s2[1];
            }
        }
        const int nq = nw0 + ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8;
        bool done = false;
        if constexpr (NF == 4) {
            // 64-column wave tiles: one more exchange (32-column half q <-> lane bit 3, a DPP row rotation by 8 under a bank mask) gives
            // every store 8 rows x 128 contiguous bytes -- whole cache lines.  Needs all 16 rows of the fragment (the lanes trade rows).
            const int row0 = m - (lane & 15);
            if (row0 + 15 < p.M) {
                u32x4 x, y;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    x[d] = (unsigned)__builtin_amdgcn_update_dpp((int)o[0][d], (int)o[1][d], 0x128, 0xf, 0xc, false);   // lanes 8-15: half 1 of rows 0-7
                    y[d] = (unsigned)__builtin_amdgcn_update_dpp((int)o[1][d], (int)o[0][d], 0x128, 0xf, 0x3, false);   // lanes 0-7: half 0 of rows 8-15
                }
                T* const dst = reinterpret_cast<T*>(outp) + (size_t)(row0 + (lane & 7)) * p.ldc + nq + ((lane >> 3) & 1) * 32;
                st_global<u32x4>(dst, x);
                st_global<u32x4>(dst + (size_t)8 * p.ldc, y);
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int q = 0; q < NF / 2; ++q) st_global<u32x4>(reinterpret_cast<T*>(outp) + (size_t)m * p.ldc + nq + q * 32, o[q]);
        }
    } else
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = nb + j * 16;
        if constexpr (EPI == EPI_STORE16 || EPI == EPI_QKV_ROPE) {
            *reinterpret_cast<v4*>(reinterpret_cast<T*>(outp) + (size_t)m * p.ldc + n) = cvt4_sat<T>(v[j]);
        } else if constexpr (EPI == EPI_STORE16_GELU) {
            const f32x4 g = gelu_erf4(v[j]);
            *reinterpret_cast<v4*>(reinterpret_cast<T*>(outp) + (size_t)m * p.ldc + n) = cvt4_sat<T>(g);
        } else if constexpr (EPI == EPI_RESID_F32) {
            f32x4* o = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outp) + (size_t)m * p.ldc + n);
            const f32x4 xn = (pre != nullptr ? pre->x[j] : ld_global<f32x4>(o)) + v[j];
            st_global<f32x4>(o, xn);
            if constexpr (LN) ln_fold_emit<T>(p, m, n, fg, xn, ln_shift);
        } else if constexpr (EPI == EPI_F32) {
            f32x4* o = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outp) + (size_t)m * p.ldc + n);
            f32x4 x = v[j];
            if (p.accumulate) {
                x += pre != nullptr && b2pre != nullptr ? pre->x[j] : ld_global<f32x4>(o);
            } else if (p.bias2 != nullptr && (p.row_period2 > 0 ? m % p.row_period2 : m) >= p.row_start2) {
                x += b2pre != nullptr ? b2pre[j] : *reinterpret_cast<const f32x4*>(p.bias2 + n);
            }
            st_global<f32x4>(o, x);
            if constexpr (LN) ln_fold_emit<T>(p, m, n, fg, x, ln_shift);
        } else if constexpr (EPI == EPI_HEAD) {
            // permuted feature n = (i*16 + jj)*7 + c ; token t of view vv at grid (gy, gx)
            const int vv = m / p.ntok, t = m - vv * p.ntok;
            const int gy = t / p.gw, gx = t - gy * p.gw;
            const int pi = n / 112, rem = n - pi * 112;
            size_t off = ((size_t)(vv * p.H + gy * 16 + pi) * p.Wimg + gx * 16) * 7 + rem;
            if (p.head_views > 0) off += (size_t)(vv / p.head_views) * (size_t)p.head_scene_skip;
            f32x4* o = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outp) + off);
            f32x4 x = v[j];
            if (p.accumulate) x += *o;
            *o = x;
        }
    }
}

// The whole wave tile (MF row fragments).  vmcnt counts loads AND stores in order, so a load issued behind a store cannot be waited for
// without waiting for the store's acknowledgement (~1 us under load): an epilogue that loads its bias / old residual / RoPE entries per row
// fragment pays one such round trip per fragment -- 8 per 128-row wave tile, ~10 us of a 256 x 256 tile whose K loop (K = 1024) takes 16 us
// (profiles/r03_gemm256k_fixed.txt: a K = 64 launch 56 us with its epilogue, 11 us without).  Here everything that depends only on the column
// (bias, bias2) is loaded once, and the row-dependent operands (old fp32 rows, positions -> table entries) BI fragments at a time, all in front
// of that batch's stores.  Values and rounding are those of epilogue_row.
template <class T, int EPI, int NF, int MF, int BI, bool LN, class LnF, bool FP = false>
__device__ __forceinline__ void epilogue_tile(const GemmArgs& p, void* const outp, const float* __restrict__ bias, const int m_first,
                                              const int nw0, const int fg, f32x4 (&acc)[MF][NF], LnF lnf, const f32x4* bpre = nullptr) {
    static_assert(!FP || (EPI == EPI_RESID_F32 && !LN && MF % 2 == 0 && NF == 4), "fold256 producer: the straight-line fp32 residual path of a 64-column wave tile");
    static_assert(MF % BI == 0, "batches of BI row fragments");
    const int nb = nw0 + fg * 4;
    const bool nobias = (EPI == EPI_F32 || EPI == EPI_HEAD) && p.accumulate;
    f32x4 b[NF], b2[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        if (bpre != nullptr) b[j] = bpre[j];   // loaded by the caller in front of its K loop (epilogue_bias): no round trip here
        else b[j] = (bias != nullptr && !nobias) ? *reinterpret_cast<const f32x4*>(bias + nb + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        b2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == EPI_F32) {
            if (!p.accumulate && p.bias2 != nullptr) b2[j] = *reinterpret_cast<const f32x4*>(p.bias2 + nb + j * 16);
        }
    }
    if constexpr (EPI == EPI_RESID_F32 && !LN && MF % 2 == 0) {
        // Whole wave tile inside the matrix (wave-uniform): straight-line code, so the compiler counts vmcnt exactly -- the old fp32 rows of the NEXT
        // pair of row fragments are requested in front of this pair's stores and waited for with those stores still in flight (vmcnt(N) waits for
        // the older loads only).  One exposed round trip per tile instead of one per batch.  Same arithmetic as epilogue_row: x + (acc + bias).
        const int lane16 = (int)__lane_id() & 15;
        if (FP || m_first - lane16 + MF * 16 - 1 < p.M) {   // (FP: launch_gemm checked M % 256 == 0)
            float* const o0 = reinterpret_cast<float*>(outp) + (size_t)m_first * p.ldc + nb;
            float sh[MF];   // FP: the rows' shifts (GemmArgs::ln_shift), requested in front of everything else
            if constexpr (FP) {
#pragma unroll
                for (int i = 0; i < MF; ++i) sh[i] = p.ln_shift != nullptr ? ld_global<float>(p.ln_shift + m_first + i * 16) : 0.f;
            }
            f32x4 x[2][2][NF];
            auto ld = [&](int pair, int slot) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < NF; ++j) x[slot][ii][j] = ld_global<f32x4>(o0 + (size_t)((2 * pair + ii) * 16) * p.ldc + j * 16);
            };
            ld(0, 0);
#pragma unroll
            for (int pair = 0; pair < MF / 2; ++pair) {
                if (pair + 1 < MF / 2) ld(pair + 1, (pair + 1) & 1);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    f32x4 xn[NF];
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        f32x4 v = acc[2 * pair + ii][j];
                        v += b[j];
                        xn[j] = x[pair & 1][ii][j] + v;
                        st_global<f32x4>(o0 + (size_t)((2 * pair + ii) * 16) * p.ldc + j * 16, xn[j]);
                    }
                    if constexpr (FP) {
                        const int m = m_first + (2 * pair + ii) * 16;
                        if (p.copy32_out != nullptr) {   // (kernel-uniform; stores only)
#pragma unroll
                            for (int j = 0; j < NF; ++j) st_global<f32x4>(p.copy32_out + (size_t)m * p.ldc + nb + j * 16, xn[j]);
                        }
                        fold256_emit<T>(p, m, nw0, fg, xn, sh[2 * pair + ii]);
                    }
                }
            }
            return;
        }
    }
    if constexpr (EPI == EPI_QKV_ROPE && !LN && MF % 2 == 0 && NF == 4 && sizeof(T) == 2) {
        // r05.  Whole wave tile inside the matrix (wave-uniform), 64-column wave tiles: straight-line code like the residual path above, so that the compiler
        // counts vmcnt exactly -- the (cos, sin) entries of the NEXT pair of row fragments are requested in front of this pair's stores and waited for with
        // those stores still in flight (the batched form below waits vmcnt(0) per batch: for the previous batch's store acknowledgements, ~1-2 us each;
        // the RoPE epilogue cost a 256 x 256 qkv tile 6.5 us more than the plain 16-bit store, profiles/r05_sparse_gemm_shapes.txt).  Rotating columns
        // only (nw0 < rope_cols, wave-uniform); the value columns take the batched path, which no longer waits per batch for loads it does not issue.
        // Same arithmetic and rounding as epilogue_row.
        const int lane = (int)__lane_id();
        const int lane16 = lane & 15;
        if (m_first - lane16 + MF * 16 - 1 < p.M && nw0 < p.rope_cols) {
            typedef typename Vec<T>::v4 v4;
            int pq[MF][2];   // clamped table row of every row fragment, per 32-column half
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int m = m_first + i * 16;
                const long long y = p.pos[(size_t)m * 2 + 0], x = p.pos[(size_t)m * 2 + 1];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    int pp = (int)((((nw0 + q * 32) >> 5) & 1) ? x : y);
                    pq[i][q] = pp < 0 ? 0 : (pp >= p.rope_npos ? p.rope_npos - 1 : pp);
                }
            }
            f32x4 t0[2][2][2], t1[2][2][2];   // [slot][row fragment of the pair][half]
            auto ld = [&](int pair, int slot) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float* tb = p.rope_tab + ((size_t)pq[2 * pair + ii][q] * 16 + fg * 4) * 2;
                        t0[slot][ii][q] = ld_global<f32x4>(tb);
                        t1[slot][ii][q] = ld_global<f32x4>(tb + 4);
                    }
            };
            const bool scale = p.out_scale != 0.f && nw0 < p.scale_cols;
            const int nq = nw0 + ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8;
            ld(0, 0);
#pragma unroll
            for (int pair = 0; pair < MF / 2; ++pair) {
                if (pair + 1 < MF / 2) ld(pair + 1, (pair + 1) & 1);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = 2 * pair + ii;
                    f32x4 v[NF];
#pragma unroll
                    for (int j = 0; j < NF; ++j) v[j] = acc[i][j] + b[j];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 a0 = t0[pair & 1][ii][q], a1 = t1[pair & 1][ii][q];
                        const float cs[4] = {a0[0], a0[2], a1[0], a1[2]};
                        const float sn[4] = {a0[1], a0[3], a1[1], a1[3]};
                        const f32x4 x0 = v[2 * q], x1 = v[2 * q + 1];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[2 * q][r] = x0[r] * cs[r] - x1[r] * sn[r];
                            v[2 * q + 1][r] = x1[r] * cs[r] + x0[r] * sn[r];
                        }
                    }
                    if (scale) {
#pragma unroll
                        for (int j = 0; j < NF; ++j) v[j] *= p.out_scale;
                    }
                    u32x4 o[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const v4 h0 = cvt4_sat<T>(v[2 * q]), h1 = cvt4_sat<T>(v[2 * q + 1]);
                        unsigned a[2], c[2];
                        __builtin_memcpy(a, &h0, 8);
                        __builtin_memcpy(c, &h1, 8);
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            auto s1 = __builtin_amdgcn_permlane32_swap(a[d], c[d], false, false);
                            auto s2 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                            o[q][d] = s2[0];
                            o[q][2 + d] = s2[1];
                        }
                    }
                    u32x4 x, y;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        x[d] = (unsigned)__builtin_amdgcn_update_dpp((int)o[0][d], (int)o[1][d], 0x128, 0xf, 0xc, false);
                        y[d] = (unsigned)__builtin_amdgcn_update_dpp((int)o[1][d], (int)o[0][d], 0x128, 0xf, 0x3, false);
                    }
                    const int row0 = m_first + i * 16 - lane16;
                    T* const dst = reinterpret_cast<T*>(outp) + (size_t)(row0 + (lane & 7)) * p.ldc + nq + ((lane >> 3) & 1) * 32;
                    st_global<u32x4>(dst, x);
                    st_global<u32x4>(dst + (size_t)8 * p.ldc, y);
                }
            }
            return;
        }
    }
    // RoPE: the positions of ALL the wave tile's rows in one go (16 bytes per row), so that a batch costs one round trip (its table entries), not two
    long long py[MF], px[MF];
    if constexpr (EPI == EPI_QKV_ROPE) {
        if (nw0 < p.rope_cols) {
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int m = m_first + i * 16;
                const int mm = m < p.M ? m : p.M - 1;
                py[i] = p.pos[(size_t)mm * 2 + 0];
                px[i] = p.pos[(size_t)mm * 2 + 1];
            }
        }
    }
#pragma unroll
    for (int i0 = 0; i0 < MF; i0 += BI) {
        EpiPre<NF> pre[BI];
        RopePre<NF> rp[BI];
#pragma unroll
        for (int ii = 0; ii < BI; ++ii) {
            const int m = m_first + (i0 + ii) * 16;
            const int mm = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                pre[ii].b[j] = b[j];
                if constexpr (EPI == EPI_RESID_F32) {
                    pre[ii].x[j] = ld_global<f32x4>(reinterpret_cast<const float*>(outp) + (size_t)mm * p.ldc + nb + j * 16);
                } else if constexpr (EPI == EPI_F32) {
                    if (p.accumulate) pre[ii].x[j] = ld_global<f32x4>(reinterpret_cast<const float*>(outp) + (size_t)mm * p.ldc + nb + j * 16);
                }
            }
        }
        // Explicit waits (the compiler's own would sit behind each fragment's row-validity branch, where only vmcnt(0) can express "the
        // loads of this batch" -- and that also waits for the previous fragment's stores): ONE wait per batch, in front of its stores.
        const bool row_loads = EPI == EPI_RESID_F32 || (EPI == EPI_F32 && p.accumulate) || (EPI == EPI_QKV_ROPE && nw0 < p.rope_cols);
        if (i0 == 0 || row_loads) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
        if constexpr (EPI == EPI_QKV_ROPE) {
            if (nw0 < p.rope_cols) {
#pragma unroll
                for (int ii = 0; ii < BI; ++ii)
#pragma unroll
                    for (int q = 0; q < NF / 2; ++q) {
                        const int nh = nw0 + q * 32;
                        int pp = (int)(((nh >> 5) & 1) ? px[i0 + ii] : py[i0 + ii]);
                        pp = pp < 0 ? 0 : (pp >= p.rope_npos ? p.rope_npos - 1 : pp);
                        const float* tb = p.rope_tab + ((size_t)pp * 16 + fg * 4) * 2;
                        rp[ii].t0[q] = *reinterpret_cast<const f32x4*>(tb);
                        rp[ii].t1[q] = *reinterpret_cast<const f32x4*>(tb + 4);
                    }
                __builtin_amdgcn_s_waitcnt(0x0f70);
            }
        }
#pragma unroll
        for (int ii = 0; ii < BI; ++ii) {
            const int m = m_first + (i0 + ii) * 16;
            if (m >= p.M) continue;
            f32x4 v[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) v[j] = acc[i0 + ii][j];
            float mu = 0.f, rstd = 0.f;
            lnf(i0 + ii, mu, rstd);
            epilogue_row<T, EPI, NF, LN>(p, outp, bias, m, nw0, fg, v, &pre[ii], mu, rstd, b2, &rp[ii]);
        }
    }
}
// the bias columns of a lane, as epilogue_tile loads them -- for callers that can afford NF x 4 registers across their K loop
template <int EPI, int NF>
__device__ __forceinline__ void epilogue_bias(const GemmArgs& p, const float* __restrict__ bias, const int nw0, const int fg, f32x4 (&b)[NF]) {
    const bool nobias = (EPI == EPI_F32 || EPI == EPI_HEAD) && p.accumulate;
#pragma unroll
    for (int j = 0; j < NF; ++j)
        b[j] = (bias != nullptr && !nobias) ? *reinterpret_cast<const f32x4*>(bias + nw0 + fg * 4 + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
}
struct NoLnFold {
    __device__ __forceinline__ void operator()(int, float&, float&) const {}
};

// WS = 2: split-weight mode, W is [N, 2K] = [W_hi | W_lo]; every K-tile stages the activation tile once plus BOTH weight
// tiles, and each activation fragment feeds two MFMAs (acc += W_hi.a ; acc += W_lo.a).
template <class T, int BM, int BN, int WGM, int WGN, int EPI, int NST, int WS, int BK, int PIPE>
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
    const int wgrp = p.wdiv > 1 ? grp / p.wdiv : grp;
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W) + (size_t)wgrp * p.strideW;
    const float* __restrict__ bias = p.bias ? p.bias + (size_t)wgrp * p.strideB : nullptr;
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
    // LN fold (consumer): row statistics of this block's BM rows, kept behind the ring
    constexpr bool LNF = BM == 64 && (EPI == EPI_STORE16 || EPI == EPI_STORE16_GELU || EPI == EPI_QKV_ROPE);   // LN-fold consumers run on the 64 x 64 tiles only
    float* const sm_ln = reinterpret_cast<float*>(smem + (size_t)NST * (BM + WS * BN) * BK * sizeof(T));
    if constexpr (LNF) {
        if (p.ln_stats != nullptr) ln_fold_rows<BM, 64 * NW>(p, m0, sm_ln, tid, n0 == 0);
    }

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

    if constexpr (PIPE) {
        // Single-round launches (one block per CU, one wave per SIMD, all waves in lock step): nothing else on the CU hides
        // the LDS-read latency of a tile, so the fragment reads are software-pipelined inside the wave -- the reads of tile
        // kt+1 are issued BEFORE the MFMAs of tile kt and retire under them.  Tile kt+1 therefore has to have landed one
        // iteration earlier than in the ring below: NST-2 younger tiles in flight instead of NST-1 (NST is 6-8 here: one
        // block per CU can spend the whole LDS on prefetch depth).  Buffer (kt-1) is overwritten by the DMA of tile
        // kt+NST-1 after the barrier of iteration kt; its reads were issued in iteration kt-2 and waited for in kt-1.
        static_assert(NST >= 4 && BK == 64, "pipelined ring");
        constexpr int IPT = PA + WS * PW;
        constexpr int PEND = (NST - 3) * IPT;       // tiles kt+2 .. kt+NST-2 may still be in flight when kt+1 is needed
        static_assert(PEND < 64, "vmcnt field");
        v8 fw[2][2][WS][NF], fa[2][2][MF];          // [set][ks]
        auto load_frags = [&](int set, int buf) {
            const T* a = sA + buf * BM * BK;
            const T* w = sW + buf * WS * BN * BK;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int lc = ks * 4 + fg;
#pragma unroll
                for (int part = 0; part < WS; ++part)
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        const int r = wn * WN + j * 16 + fr;
                        fw[set][ks][part][j] = *reinterpret_cast<const v8*>(w + (part * BN + r) * BK + swzk<BK>(r, lc) * 8);
                    }
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    const int r = wm * WM + i * 16 + fr;
                    fa[set][ks][i] = *reinterpret_cast<const v8*>(a + r * BK + swzk<BK>(r, lc) * 8);
                }
            }
        };
        auto wait_dma = [&](int younger) {          // younger = tiles issued after the one that must have landed
            if (younger >= NST - 3) __builtin_amdgcn_s_waitcnt(0x0f70 | (PEND & 15) | ((PEND >> 4) << 14));
            else __builtin_amdgcn_s_waitcnt(0x0f70);   // tail: drain
        };
#pragma unroll
        for (int t = 0; t < NST - 1; ++t)
            if (t < nk) stage(t, t);
        // tile 0: everything issued so far except tiles 1..NST-2 must be in
        if (nk - 1 >= NST - 2) __builtin_amdgcn_s_waitcnt(0x0f70 | (((NST - 2) * IPT) & 15) | ((((NST - 2) * IPT) >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
        load_frags(0, 0);
        int buf = 0;
        // two iterations per trip so that the fragment set index is a compile-time constant (registers, not scratch)
#ifdef GEMM_TRACE
#define M3R_STAMP(slot) do { if (blockIdx.x == 0 && tid == 0 && kt < 64) g_gemm_trace[kt * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define M3R_STAMP(slot) do { } while (0)
#endif
        auto body = [&](auto curc, int kt) {
            constexpr int cur = decltype(curc)::value;
            const int nbuf = buf + 1 == NST ? 0 : buf + 1;
            M3R_STAMP(0);
            if (kt + 1 < nk) wait_dma(nk - 2 - kt < NST - 3 ? nk - 2 - kt : NST - 3);
            M3R_STAMP(1);
            __builtin_amdgcn_s_barrier();
            M3R_STAMP(2);
            const int nt = kt + NST - 1;
            if (nt < nk) {
                int sb = buf + NST - 1;
                sb = sb >= NST ? sb - NST : sb;
                stage(nt, sb);                          // buffer of tile kt-1
            }
            M3R_STAMP(3);
            // Fragments of tile kt were read one iteration ago.  "Using" them here makes the compiler place its own wait for
            // them BEFORE the next tile's reads are issued; with an opaque inline s_waitcnt it keeps treating the
            // loop-carried registers as pending and drains the new reads as well (lgkmcnt(0)) in front of the MFMAs.
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int part = 0; part < WS; ++part)
#pragma unroll
                    for (int j = 0; j < NF; ++j) asm volatile("" ::"v"(fw[cur][ks][part][j]));
#pragma unroll
                for (int i = 0; i < MF; ++i) asm volatile("" ::"v"(fa[cur][ks][i]));
            }
            M3R_STAMP(4);
            if (kt + 1 < nk) load_frags(cur ^ 1, nbuf);
            M3R_STAMP(5);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int part = 0; part < WS; ++part)
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int j = 0; j < NF; ++j) acc[i][j] = mfma16(fw[cur][ks][part][j], fa[cur][ks][i], acc[i][j]);
            M3R_STAMP(6);
            buf = nbuf;
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            body(std::integral_constant<int, 0>{}, kt);
            body(std::integral_constant<int, 1>{}, kt + 1);
        }
        if (kt < nk) body(std::integral_constant<int, 0>{}, kt);
    } else if constexpr (NST == 2) {
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
    auto lnf = [&](int i, float& mu, float& rstd) {
        if constexpr (LNF) {
            if (p.ln_stats != nullptr) {
                constexpr int TPR = 64 * NW / BM;
                mu = sm_ln[BM * TPR * 2 + (wm * WM + i * 16 + fr) * 2];
                rstd = sm_ln[BM * TPR * 2 + (wm * WM + i * 16 + fr) * 2 + 1];
            }
        }
    };
    epilogue_tile<T, EPI, NF, MF, (MF % 4 == 0 ? 4 : (MF % 2 == 0 ? 2 : 1)), BM == 64>(p, outp, bias, m0 + wm * WM + fr, n0 + wn * WN, fg, acc, lnf);
}

template <class T, int BM, int BN, int WGM, int WGN, int EPI, int NST, int WS, int BK = 64, int PIPE = 0>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
    const int nbn = a.N / BN, nbm = (a.M + BM - 1) / BM;
    const size_t lds = (size_t)NST * (BM + WS * BN) * BK * sizeof(T) + ln_fold_lds_bytes<BM, 64 * WGM * WGN>();
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, BM, BN, WGM, WGN, EPI, NST, WS, BK, PIPE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WGM, WGN, EPI, NST, WS, BK, PIPE>), dim3(nbm * nbn, a.batch > 1 ? a.batch : 1),
                       dim3(64 * WGM * WGN), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------------------------
// 8-wave, one-block-per-CU GEMM for chip-filling launches: 256 x 256 tile (plain weights) / 256 x 128 hi+lo (split),
// K-tile 32, 4-stage LDS ring (128 KB), two wave groups staggered by one barrier.
//
// Waves 0-3 (group 0, rows 0-127 of the tile) and 4-7 (group 1, rows 128-255) share the 4 SIMDs pairwise.  Every K-tile
// costs a wave two block barriers: [wait DMA] B [12 ds_read_b128 + issue the DMA of tile t+3] [lgkmcnt 0] B [32 MFMA].
// Group 1 runs exactly one barrier behind group 0, so in every barrier interval one wave of each SIMD streams its 32
// MFMAs (s_setprio 1) while its partner does the LDS reads and DMA issue for its next tile: the matrix pipe sees
// back-to-back MFMAs instead of the read->multiply serialisation of the 4-wave kernels above.
//   RAW  every wave waits (counted vmcnt, 2 younger tiles stay in flight) for ITS pieces of tile t before global
//        barrier #2t; group 0 reads the tile after #2t, group 1 after #2t+1.
//   WAR  tile t+3 lands in the buffer of tile t-1.  Both groups retire their ds_reads (lgkmcnt 0) BEFORE the barrier
//        that ends their read interval, so after #2t nobody has a read of tile t-1 in flight; group 0 issues the DMA
//        in (#2t, #2t+1), group 1 in (#2t+1, #2t+2).
// The accumulation order per output (k ascending, hi before lo inside each 32-deep step) is the same as in gemm_kernel,
// so the two kernels produce identical bits.
// OCC = 2 (experiment, split BN = 128 only): a 2-slot ring (64 KB) and at most 128 VGPRs, so that TWO blocks share a CU -- one block's
// epilogue / prologue then runs under the other's K loop, and 16 waves per CU overlap the three pipes better (profiles/r02_pipe_rates.txt).
template <class T, int EPI, int WS, int BN, int OCC = 1>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(OCC == 2 ? 4 : 2, OCC == 2 ? 4 : 2))) gemm256_kernel(const GemmArgs p) {
    typedef typename Vec<T>::v8 v8;
    constexpr int BM = 256, BK = 32;
    constexpr int WR = WS * BN;                // rows of the staged weight region: [hi BN rows | lo BN rows] when split
    constexpr int NST = OCC == 2 ? 2 : (WR >= 384 ? 3 : 4);     // 48 / 40 KB stages x 3 (split, BN 256 / 192) or 32 KB stages x 4
    constexpr int WN = BN / 4;                 // wave tile: 128 (m) x WN (n)
    constexpr int MF = 8, NF = WN / 16;        // 16x16 fragments per wave tile
    constexpr int PW = WR / 128;               // weight DMA instructions per wave and K-tile (A: 2)
    constexpr int IPT = 2 + PW;
    constexpr int STAGE = (BM + WR) * BK;      // elements per stage: A [256][32] then W [WR][32]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* const lds = reinterpret_cast<T*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nbn = p.N / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    constexpr int GM = 4;                      // grouped order: 4 row-blocks x all column-blocks per group
    const int tpg = GM * nbn;
    const int gidx = bid / tpg;
    const int gfirst = gidx * GM;
    const int gsz = (nbm - gfirst < GM) ? nbm - gfirst : GM;
    const int gin = bid - gidx * tpg;
    const int m0 = (gfirst + gin % gsz) * BM;
    const int n0 = (gin / gsz) * BN;

    const int grp = blockIdx.y;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + (size_t)grp * p.strideA;
    const int wgrp = p.wdiv > 1 ? grp / p.wdiv : grp;
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W) + (size_t)wgrp * p.strideW;
    const float* __restrict__ bias = p.bias ? p.bias + (size_t)wgrp * p.strideB : nullptr;
    void* const outp = p.out_table ? p.out_table[grp] : p.out;

    // ---- staging: one wave instruction moves 16 rows x 64 B; per K-tile a wave issues 2 (A) + 2 (W) instructions
    const int srow = lane >> 2, pch = lane & 3;
    const T* a_src[2];
    const T* w_src[PW];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r = (wave * 2 + t) * 16 + srow;            // 0..255
        int gr = m0 + r;
        gr = gr < p.M ? gr : p.M - 1;
        a_src[t] = A + (size_t)gr * p.lda + swz32(r, pch) * 8;
    }
#pragma unroll
    for (int t = 0; t < PW; ++t) {
        // weight region row r: rows [0, BN) = (hi) weight rows n0 .. n0 + BN, rows [BN, 2 BN) = their lo halves
        const int r = (wave * PW + t) * 16 + srow;           // 0..WR-1
        const int part = r / BN, wrow = r - part * BN;
        w_src[t] = W + (size_t)(n0 + wrow) * (size_t)(p.K * WS) + (size_t)part * p.K + swz32(r, pch) * 8;
    }
    auto stage = [&](int kt, int buf) {
        T* base = lds + buf * STAGE;
#pragma unroll
        for (int t = 0; t < 2; ++t) glds16(a_src[t] + kt * BK, base + (wave * 2 + t) * 16 * BK);
#pragma unroll
        for (int t = 0; t < PW; ++t) glds16(w_src[t] + kt * BK, base + BM * BK + (wave * PW + t) * 16 * BK);
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int nk = p.K / BK;
    // LDS element offsets of this lane's fragments inside a stage
    int a_off[MF], w_off[WS][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int r = wr * 128 + i * 16 + fr;
        a_off[i] = r * BK + swz32(r, fg) * 8;
    }
#pragma unroll
    for (int part = 0; part < WS; ++part)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int r = part * BN + wc * WN + j * 16 + fr;
            w_off[part][j] = BM * BK + r * BK + swz32(r, fg) * 8;
        }

    // wait until this wave's DMA pieces of tile u have landed: at most min(NST-2, nk-1-u) younger tiles (IPT loads each) pending
    auto wait_tile = [&](int u) {
        const int younger = nk - 1 - u;
        if (NST >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPT) : "memory");
        else if (NST >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) stage(t, t);
    if (wr == 1) {
        wait_tile(0);
        __builtin_amdgcn_s_barrier();
    }
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
        M3R_STAMP256(0);
#if defined(G256_EXP_NOLADDER)   // probe builds only (timing of the hand-over; the last K-tiles would race): one unconditional counted wait
        if (wr == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST >= 4 ? 2 * IPT : IPT) : "memory");
#else
        if (wr == 0) wait_tile(t);
#endif
        M3R_STAMP256(1);
        __builtin_amdgcn_s_barrier();
        M3R_STAMP256(2);
        // ---- read interval
        const T* base = lds + buf * STAGE;
        v8 af[MF], wf[WS][NF];
#pragma unroll
        for (int part = 0; part < WS; ++part)
#pragma unroll
            for (int j = 0; j < NF; ++j) wf[part][j] = *reinterpret_cast<const v8*>(base + w_off[part][j]);
#pragma unroll
        for (int i = 0; i < MF; ++i) af[i] = *reinterpret_cast<const v8*>(base + a_off[i]);
        if (t + NST - 1 < nk) {
            int nb = buf + NST - 1;
            nb = nb >= NST ? nb - NST : nb;
            stage(t + NST - 1, nb);
        }
        M3R_STAMP256(3);
        if (wr == 1 && t + 1 < nk) wait_tile(t + 1);
        M3R_STAMP256(4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        M3R_STAMP256(5);
        __builtin_amdgcn_s_barrier();
        M3R_STAMP256(6);
        // ---- multiply interval
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int part = 0; part < WS; ++part)
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = mfma16(wf[part][j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        M3R_STAMP256(7);
        buf = buf + 1 == NST ? 0 : buf + 1;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();

    epilogue_tile<T, EPI, NF, MF, 4, false>(p, outp, bias, m0 + wr * 128 + fr, n0 + wc * WN, fg, acc, NoLnFold{});
}

// (r03, built / measured / removed: `gemm256s_kernel` -- the same tile, ring and epilogues with ALL EIGHT waves multiplying all the time,
// every wave streaming its own fragments under its own MFMAs (activation fragments two row fragments ahead, the next tile's weight fragments
// behind the last two, hand-counted lgkmcnt from inline-asm reads, ONE block barrier per K-tile, no hand-over between wave groups).  Same bits;
// nine chip-filling shapes: 907 us against 913 us on plain fp16 weights, 1382 against 1323 us on split weights (profiles/r03_gemm256s_ab.txt).
// Two loop structures this different landing on the same time says the hand-over hole is not the limit: a K-tile costs a SIMD the SUM of what
// its two waves issue -- 2 x (32 MFMAs x 16 + 4 LDS-DMA pieces x ~64 + 12 ds_read_b128 x ~27 cycles) = ~2200 cycles against 1024 of MFMA time
// (plain weights; measured ~2000) -- whichever wave issues what when.  Only fewer non-MFMA issue cycles per MFMA move it: weight fragments
// straight from L2 into registers (no DMA, no ds_read for them), or 128 x 128 wave tiles.  The kernel is in the git history
// (commit "gemm256s_kernel: streamed-fragment 256-row GEMM"); scripts/checks/asm_inflight_regs.py, written for it, stays.)
template <class T, int EPI, int WS, int BN, int OCC = 1>
static int launch_256(const GemmArgs& a, hipStream_t s) {
    const int nbn = a.N / BN, nbm = (a.M + 255) / 256;
    const size_t lds = (size_t)(OCC == 2 ? 2 : (WS * BN >= 384 ? 3 : 4)) * (256 + WS * BN) * 32 * sizeof(T);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<T, EPI, WS, BN, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm256_kernel<T, EPI, WS, BN, OCC>), dim3(nbm * nbn, a.batch > 1 ? a.batch : 1), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------------------------
// gemm256k_kernel (r03): 256 x BN tile, 8 waves, K staged in CHUNKS of 64 (128-byte operand rows), two 64 KB buffers, ONE barrier per chunk.
//   Why (profiles/r03_pipe_rates.txt): an LDS-DMA instruction that gathers 16 rows x 64 B (the K-tile of 32 of gemm256_kernel) touches 16
//   cache lines for 1 KB; the CU's vector-memory front end delivers such pieces at 30 B/clk -- 1070 cycles for the 32 KB of a 256 x 256 x 32
//   K-tile, as long as the tile's 1024 cycles of MFMA work, and a wave that cannot issue its DMA cannot issue its MFMAs either.  Pieces of
//   8 rows x 128 B (whole lines) go at 59 B/clk.  A chunk is two 32-deep sub-tiles: 16 phases of NF x WS MFMAs per wave; the fragments
//   are streamed from inline-asm ds_reads with hand-counted lgkmcnt (activation fragments two phases ahead in a ring of four register
//   sets, the weight fragments of the next sub-tile behind the last two phases of the current one).
//   The chunk barrier sits in front of phase 14: it publishes chunk c+1 (every wave has waited for ITS pieces: vmcnt 0) and, because
//   every wave has also retired its last read of chunk c (lgkmcnt 0 on the two fragments still in flight), frees the buffer of chunk c for
//   chunk c+2, whose DMA pieces are issued one per phase over the next NP phases.  (Measured and dropped, profiles/r03_gemm256k_ab.txt:
//   spreading them over all 16 phases, even / odd waves alternating: -9 %, the last pieces land too late; one dword per lane touching the
//   cache lines of chunk c+3 a chunk ahead of its DMA: -12 %.  Staggering the two waves of a SIMD (w, w + 4) so that they never issue DMA in
//   the same phase -- the ablations say the DMA issue costs 25 % of the loop, the waves blocked on the vector-memory front end together --
//   as run-time phase tests: -55 % (128 scalar branches per chunk break the MFMA stream, profiles/r03_gemm256k_spread.txt); as two copies of
//   the phase loop chosen per wave group: the register allocator copies in-flight fragments at the join (scripts/checks/asm_inflight_regs.py
//   rejects the build).)
// Same accumulation order per output as every other tile shape (k ascending, hi before lo in each 32-deep step): identical bits.
template <class T, int EPI, int WS, int BN, int ABL = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) gemm256k_kernel(const GemmArgs p) {
    typedef typename Vec<T>::v8 v8;
    constexpr int BM = 256, CK = 64;
    constexpr int WR = WS * BN;                 // rows of the staged weight region
    static_assert(WR == 256 || WR == 128, "two chunk buffers must fit the LDS");
    constexpr int WN = BN / 4;
    constexpr int MF = 8, NF = WN / 16;
    constexpr int NW = NF * WS, NW2 = NW / 2;   // weight fragments per sub-tile
    static_assert(NW >= 2 && NW % 2 == 0, "weight fragments are read in two halves");
    constexpr int PW = WR / 64;                 // weight DMA pieces (8 rows x 128 B) per wave and chunk; activations: 4
    constexpr int NP = 4 + PW;
    constexpr int STAGE = (BM + WR) * CK;       // elements per buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* const lds = reinterpret_cast<T*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nbn = p.N / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    constexpr int GM = 4;
    const int tpg = GM * nbn;
    const int gidx = bid / tpg;
    const int gfirst = gidx * GM;
    const int gsz = (nbm - gfirst < GM) ? nbm - gfirst : GM;
    const int gin = bid - gidx * tpg;
    const int m0 = (gfirst + gin % gsz) * BM;
    const int n0 = (gin / gsz) * BN;

    const int grp = blockIdx.y;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + (size_t)grp * p.strideA;
    const int wgrp = p.wdiv > 1 ? grp / p.wdiv : grp;
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W) + (size_t)wgrp * p.strideW;
    const float* __restrict__ bias = p.bias ? p.bias + (size_t)wgrp * p.strideB : nullptr;
    void* const outp = p.out_table ? p.out_table[grp] : p.out;

    // ---- staging: one wave instruction moves 8 rows x 128 B
    const int srow = lane >> 3, pch = lane & 7;
    const T* a_src[4];
    const T* w_src[PW];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = (wave * 4 + t) * 8 + srow;
        int gr = m0 + r;
        gr = gr < p.M ? gr : p.M - 1;
        a_src[t] = A + (size_t)gr * p.lda + swz(r, pch) * 8;
    }
#pragma unroll
    for (int t = 0; t < PW; ++t) {
        const int r = (wave * PW + t) * 8 + srow;            // weight region row: [hi BN rows | lo BN rows] when split
        const int part = r / BN, wrow = r - part * BN;
        w_src[t] = W + (size_t)(n0 + wrow) * (size_t)(p.K * WS) + (size_t)part * p.K + swz(r, pch) * 8;
    }
    auto piece = [&](int q, int c, int buf) {   // q: compile-time constant at every call site
        T* base = lds + buf * STAGE;
        if (q < 4) glds16(a_src[q < 4 ? q : 0] + c * CK, base + (wave * 4 + q) * 8 * CK);
        else glds16(w_src[q >= 4 ? q - 4 : 0] + c * CK, base + BM * CK + (wave * PW + (q - 4)) * 8 * CK);
    };
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int nc = p.K / CK;
    // LDS byte addresses of this lane's fragments in buffer 0, sub-tile h: row-fragment i / weight fragment (part, j) are immediates on top
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
    unsigned a_lane[2], w_lane[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ra = wr * 128 + fr, rw = wc * WN + fr;
        a_lane[h] = lds0 + (unsigned)(ra * 128 + swz(ra, h * 4 + fg) * 16);
        w_lane[h] = lds0 + (unsigned)(BM * 128 + rw * 128 + swz(rw, h * 4 + fg) * 16);
    }
    constexpr unsigned STAGEB = STAGE * 2;
    f32x4 bpre[NF];   // bias columns: loaded here, in front of the first DMA, instead of as a round trip between the K loop and the stores
    epilogue_bias<EPI, NF>(p, bias, n0 + wc * WN, fg, bpre);

#pragma unroll
    for (int q = 0; q < NP; ++q) piece(q, 0, 0);
    if (nc > 1) {
#pragma unroll
        for (int q = 0; q < NP; ++q) piece(q, 1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    v8 wf[2][WS][NF], af[4];
#define M3R_DSR0(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define M3R_DSR(dst, addr, off) do { if constexpr (!(ABL & 2)) M3R_DSR0(dst, addr, off); else asm volatile("" : "+v"(dst)); } while (0)
#define M3R_LGKM(n, x) do { if constexpr (!(ABL & 2)) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(n)); } while (0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // scalar loads share the counter: none may be in flight under the counted waits
#pragma unroll
    for (int part = 0; part < WS; ++part)
#pragma unroll
        for (int j = 0; j < NF; ++j) M3R_DSR0(wf[0][part][j], w_lane[0], (part * BN + j * 16) * 128);
    M3R_DSR0(af[0], a_lane[0], 0);
    M3R_DSR0(af[1], a_lane[0], 16 * 128);
    if constexpr ((ABL & 2) != 0) {   // timing ablation: fragments read once
        M3R_DSR0(af[2], a_lane[0], 32 * 128);
        M3R_DSR0(af[3], a_lane[0], 48 * 128);
#pragma unroll
        for (int part = 0; part < WS; ++part)
#pragma unroll
            for (int j = 0; j < NF; ++j) M3R_DSR0(wf[1][part][j], w_lane[1], (part * BN + j * 16) * 128);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    for (int c = 0; c < nc; ++c) {
        const int buf = c & 1;
        const unsigned cur = buf ? STAGEB : 0u, nxt = buf ? 0u : STAGEB;
        const unsigned a_cur0 = a_lane[0] + cur, a_cur1 = a_lane[1] + cur, a_nxt0 = a_lane[0] + nxt;
        const unsigned w_cur1 = w_lane[1] + cur, w_nxt0 = w_lane[0] + nxt;
        const bool dma_next = c >= 1 && c + 1 < nc;   // the rest of chunk c+1 (its first pieces went out behind the barrier of chunk c-1)
        const bool dma_next2 = c + 2 < nc;            // chunk c+2, behind this chunk's barrier
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int h = g >> 3, i = g & 7;
            if (g == 14) {
                // last reads of this chunk (fragments 14, 15) retired; own pieces of chunk c+1 landed; then: chunk c+1 visible, this buffer free
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[2]), "+v"(af[3]));
                if constexpr (!(ABL & 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
            }
            // ---- reads: activation fragment g+2 (phases 14 / 15: fragments 0 / 1 of chunk c+1 -- the last chunk reads stale ring contents
            // nobody uses, so that the counted waits are the same in every chunk and no branch surrounds a statement with an in-flight register)
            {
                const int g2 = g + 2;
                if (g2 < 8) M3R_DSR(af[g2 & 3], a_cur0, g2 * 16 * 128);
                else if (g2 < 16) M3R_DSR(af[g2 & 3], a_cur1, (g2 - 8) * 16 * 128);
                else M3R_DSR(af[g2 & 3], a_nxt0, (g2 - 16) * 16 * 128);
            }
            if (i >= 6) {   // the next sub-tile's weight fragments: half of them behind each of the last two phases
#pragma unroll
                for (int part = 0; part < WS; ++part)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        if ((part * NF + j) / NW2 == i - 6) {
                            if (h == 0) M3R_DSR(wf[1][part][j], w_cur1, (part * BN + j * 16) * 128);
                            else M3R_DSR(wf[0][part][j], w_nxt0, (part * BN + j * 16) * 128);
                        }
            }
            // ---- DMA: phase offset o behind the barrier -> piece o
            {
                const int o = (g + 2) & 15;
                const bool on = (ABL & 1) ? false : (g >= 14 ? dma_next2 : dma_next);
                const int tc = g >= 14 ? c + 2 : c + 1, tb = g >= 14 ? buf : (buf ^ 1);
                if (o < NP && on) piece(o < NP ? o : 0, tc, tb);
            }
            // ---- wait for fragment g (phase 0 of a sub-tile: and for its weight fragments)
            if (g >= 14) {
            } else if (i == 0) {
                M3R_LGKM(1, af[g & 3]);
#pragma unroll
                for (int part = 0; part < WS; ++part)
#pragma unroll
                    for (int j = 0; j < NF; ++j) asm volatile("" : "+v"(wf[h][part][j]));
            } else if (i < 6) {
                M3R_LGKM(2, af[g & 3]);
            } else if (i == 6) {
                M3R_LGKM(2 + NW2, af[g & 3]);
            } else {
                M3R_LGKM(2 + 2 * NW2, af[g & 3]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int part = 0; part < WS; ++part)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    if constexpr (!(ABL & 8)) acc[i][j] = mfma16(wf[h][part][j], af[g & 3], acc[i][j]);
                    else asm volatile("" : "+v"(acc[i][j]) : "v"(wf[h][part][j]), "v"(af[g & 3]));
                }
            __builtin_amdgcn_s_setprio(0);
        }
    }
    // the last chunk's "next chunk" reads are still in flight: their destination registers stay LIVE up to this wait, or the epilogue
    // would reuse them under the landing data (scripts/checks/asm_inflight_regs.py walks the generated code for exactly that)
    if constexpr (NW == 4 && WS == 1)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(wf[0][0][0]), "+v"(wf[0][0][1]), "+v"(wf[0][0][2]), "+v"(wf[0][0][3]) : : "memory");
    else if constexpr (NW == 4 && WS == 2)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(wf[0][0][0]), "+v"(wf[0][0][1]), "+v"(wf[0][1][0]), "+v"(wf[0][1][1]) : : "memory");
    else {
        static_assert(NW == 2 && WS == 1, "gemm256k: add the drain statement for this geometry");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(wf[0][0][0]), "+v"(wf[0][0][1]) : : "memory");
    }
#undef M3R_DSR
#undef M3R_DSR0
#undef M3R_LGKM

    if constexpr ((ABL & 32) != 0) {   // timing ablation: no epilogue
        if (acc[0][0][0] != 12345.678f) return;
    }
    epilogue_tile<T, EPI, NF, MF, 4, false>(p, outp, bias, m0 + wr * 128 + fr, n0 + wc * WN, fg, acc, NoLnFold{}, bpre);
}

template <class T, int EPI, int WS, int BN, int ABL = 0>
static int launch_256k_v(const GemmArgs& a, hipStream_t s) {
    const int nbn = a.N / BN, nbm = (a.M + 255) / 256;
    const size_t lds = (size_t)2 * (256 + WS * BN) * 64 * sizeof(T);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256k_kernel<T, EPI, WS, BN, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm256k_kernel<T, EPI, WS, BN, ABL>), dim3(nbm * nbn, a.batch > 1 ? a.batch : 1), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
template <class T, int EPI, int WS, int BN>
static int launch_256k(const GemmArgs& a, hipStream_t s) {
#ifdef M3R_GEMM_EXPERIMENTS
    // timing ablations (wrong results by construction): M3R_G256K_ABL bit 0 no DMA, 1 no fragment reads, 2 no barrier, 3 no MFMA, 4 no vmcnt wait, 5 no epilogue
    if constexpr (EPI == EPI_STORE16 && WS == 1) {
        static int abl = -1;
        if (abl < 0) {
            const char* e = getenv("M3R_G256K_ABL");
            abl = e ? atoi(e) : 0;
        }
        switch (abl) {
            case 0: break;
#define M3R_ABL(n) case n: return launch_256k_v<T, EPI, WS, BN, n>(a, s);
            M3R_ABL(1) M3R_ABL(2) M3R_ABL(3) M3R_ABL(4) M3R_ABL(5) M3R_ABL(7) M3R_ABL(8) M3R_ABL(9) M3R_ABL(12) M3R_ABL(20) M3R_ABL(32) M3R_ABL(39)
#undef M3R_ABL
            default: return 1;
        }
    }
#endif
    return launch_256k_v<T, EPI, WS, BN>(a, s);
}

// ------------------------------------------------------------------------------------------------------------------
// gemm256p_kernel (r05): 256 x BN tile, 8 waves, K-tiles of 64 (128-byte operand rows like gemm256k), the two wave groups ALTERNATING between a
// load interval and a multiply interval (like gemm256_kernel), and the tile staged in HALF-TILES of 16 KB that are restaged one phase after their
// last read -- the "8-phase" structure of cdna_hip_programming.md section 5, which that guide measures at 1320-1470 TF/s on random operands where the
// symmetric loop of gemm256k_kernel reaches ~1050 (K loop only, profiles/r03_gemm256k_fixed2.txt).
//   Why the symmetric loop stalls: its eight waves run the same instruction stream in near lock step behind one barrier per chunk, so both waves of a
//   SIMD reach their LDS-DMA instruction together, both sit in the CU's vector-memory front end (8 pieces x ~17 cycles), and the matrix pipe idles
//   until the first of them gets back to its MFMAs (MFMA busy 0.46-0.48, profiles/r04_pmc_sq.json).  Here a wave issues NO memory instruction between
//   the first and the last MFMA of a multiply interval, and while it multiplies its SIMD partner (the other group) does all of its reads and DMA.
// Geometry.  Waves (wr, wc) = (wave >> 2, wave & 3): wave tile 128 rows x BN/4 columns.  Per K-tile four half-tiles of 128 LDS rows x 128 B:
//   A_h (h = 0, 1): rows {wr' * 128 + h * 64 + r} of the tile at LDS row wr' * 64 + r  -- every wave's row sub-tile h (4 row fragments)
//   B_g (g = 0, 1): plain weights: weight rows {wc' * 64 + g * 32 + r} at LDS row wc' * 32 + r -- every wave's column sub-tile g (2 fragments);
//                   split weights (BN = 128): part g (hi / lo) of weight rows {wc' * 32 + r}.
//   Two buffers of 4 x 16 KB.  A phase = [load interval | barrier | multiply interval | barrier]; group 1 runs one barrier behind group 0.
//   NPH = 4 (plain): phases (a0,b0) (a0,b1) (a1,b1) (a1,b0), 16 MFMAs each; reads 12 / 4 / 8 / 0 ds_read_b128; ONE half-tile (2 DMA pieces per wave) per phase.
//   NPH = 2        : phases a0 x (b0,b1), a1 x (b0,b1), 32 MFMAs each (split: hi then lo inside each 32-deep step); reads 16 / 8; two half-tiles per phase.
// Ordering (cdna_hip_programming.md "Read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  a wave's counted vmcnt for a half-tile sits in the load interval of phase w (group 0: before global barrier #2w, group 1: before #2w+1);
//        the first read of that half-tile is in phase w+1 (group 0: after #2w+1, group 1: after #2w+2).
//   WAR  every wave retires its ds_reads (lgkmcnt 0) BEFORE the barrier that ends its load interval, so a half-tile last read in phase r is
//        restaged in phase r+1 at the earliest: group 0 issues after #2r+1 (group 1's reads retired before #2r+1), group 1 after #2r+2.
// Same accumulation order per output as every other tile shape (k ascending, hi before lo in each 32-deep step): identical bits.
// SYNC = 2: two barriers per phase, the groups' multiply intervals exclusive (above).  SYNC = 1 (NPH = 2 only): ONE barrier per phase.  Between two
// barriers group 0 runs [load block of phase p | its 32 MFMAs of phase p] and group 1 [its 32 MFMAs of phase p-1 | load block of phase p]: the two waves of
// a SIMD are in anti-phase by construction, but nothing stops the wave that finishes its loads early from multiplying beside its partner (two waves
// alternating MFMAs issue one every 16 cycles where a single wave needs ~17.8, profiles/r02_gemm256_trace.txt), and the hand-over hole of the exclusive
// form (~80 cycles per barrier with the matrix pipe drained, same trace) is paid once per phase instead of twice.
//   RAW  the counted wait for phase q's half-tiles sits in every wave's load block of interval q-1 (before barrier #q); both groups read them in interval q.
//   WAR  group 1 retires its reads of phase r at the end of interval r (lgkmcnt 0 before barrier #r+1); the restage goes out in interval r+1.
// PERS = 1 (r06): PERSISTENT tiles.  The grid is one block per CU (launch_256p) and a block walks the work items blockIdx.x, + gridDim.x, ... (work item = tile of
// problem `grp`); the block -> tile map per item is the non-persistent one, so a CU's sequence of tiles is what the dispatcher's round-robin gives the plain launch.
// What it removes is the tile boundary of a multi-round launch: s_endpgm waits for the wave's outstanding stores, the workgroup's 160 KB of LDS are released only when its
// last wave has ended, the next workgroup is dispatched, reads its kernel arguments and only then issues its first DMA (~3000 cycles to the first MFMA) -- here the
// stores of tile i drain under the prologue of tile i + 1, issued straight behind them (the vendor's stream-K kernel has the same loop: profiles/r06_vendor_tile_boundary.txt).
// LDS across the boundary: after the barrier that re-aligns the two wave groups at the end of the K loop every wave has retired its last fragment reads (lgkmcnt(0) in front of
// its last load-interval barrier), the epilogue does not touch LDS, so the next prologue's DMA may start in any wave at once.  Same accumulation order: same bits.
template <class T, int EPI, int WS, int BN, int NPH, int SYNC = 2, int PERS = 0, int FOLD = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) gemm256p_kernel(const GemmArgs p) {
    typedef typename Vec<T>::v8 v8;
    constexpr int BM = 256, BK = 64;
    // FOLD (r06): LayerNorm folded into the chip-filling launches (GemmArgs::fold256) -- consumer for the 16-bit-store epilogues, producer for the fp32 residual one
    constexpr bool FOLDC = FOLD != 0 && (EPI == EPI_STORE16 || EPI == EPI_STORE16_GELU || EPI == EPI_QKV_ROPE);
    constexpr bool FOLDP = FOLD != 0 && EPI == EPI_RESID_F32;
    static_assert(FOLD == 0 || ((FOLDC || FOLDP) && PERS == 0 && SYNC == 2 && BN == 256 && sizeof(T) == 2), "fold256: 64-column wave tiles, one tile per block");
    static_assert((WS <= 2 && WS * BN == 256) || (WS == 3 && BN == 128), "the staged weight region is 256 rows: 256 plain columns or 128 columns hi + lo");
    static_assert(NPH == 2 || (NPH == 4 && WS == 1), "the four-phase form multiplies (a, b0) and (a, b1) in different phases: plain weights only");
    static_assert(WS != 3 || (NPH == 2 && SYNC == 2 && sizeof(T) == 2 && !__is_same(T, bf16_t)), "sparse low part: fp16, two-phase two-barrier form");
    // WS = 3 (r05): split weights whose LOW part is 2:4-sparse -- in every group of 4 consecutive k of a weight row only the 2 entries of largest magnitude
    // are kept (packed at finalize, misc.hip::sparse24_pack_kernel) and the low product of a whole 64-deep K-tile is ONE v_smfmac_f32_16x16x64_f16 per output
    // fragment (the sparse matrix is the MFMA's A operand = our weight tile; its dense B operand is the two 32-deep activation fragments the hi product
    // already holds, side by side; profiles/r05_smfmac_probe.txt: operand layout and 2x the dense rate) instead of two dense MFMAs: 48 matrix instructions per
    // wave and K-tile where the dense split issues 64.  What the dropped (smaller) half of W_lo costs was emulated first (scripts/emul/sparse_lo.py:
    // fp16wa 6.05e-4 / 5.68e-4 dense, 5.99e-4 / 5.64e-4 sparse: the Mlp Linears' plain weights carry the error, not what is left of the low parts).
    // Half-tile B1 then holds [128 rows][32 kept values] (8 KB; K-tile-major in memory: one contiguous KB per wave) and, behind them, each wave's own copy of its
    // 256 bytes of 2-bit positions (one dword per lane: low half fragment 0, high half fragment 1 = the instruction's ABID).
    static_assert(SYNC == 2 || NPH == 2, "the one-barrier form is built on the two-phase schedule");
    constexpr int WN = BN / 4, MF = 8, NF = WN / 16;       // plain: NF = 4 (two column sub-tiles of 2); split: NF = 2 (hi and lo fragments of the same 2)
    constexpr unsigned HALFB = 128 * BK * 2;               // bytes per half-tile (16 KB)
    constexpr unsigned BUFB = 4 * HALFB;                   // A0 A1 B0 B1
    extern __shared__ __attribute__((aligned(16))) char smem[];

    for (int work = blockIdx.x;; work += gridDim.x) {   // PERS = 0: one pass (the body is not re-indented)
    int tid_ = threadIdx.x;
    // PERS: everything derived from the lane is recomputed per tile (a few integer instructions) -- hoisted out of the tile loop it would have to live across the
    // epilogue, and the kernel sits at the 256-register cap of two waves per SIMD (spills)
    if constexpr (PERS != 0) asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nbn = p.N / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;
    const int nwork = PERS ? nwg * (p.batch > 1 ? p.batch : 1) : 0;
    if constexpr (PERS != 0) { if (work >= nwork) break; }
    const int grp = PERS ? work / nwg : (int)blockIdx.y;
    int bid = PERS ? work - grp * nwg : (int)blockIdx.x;
    {
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int GM = p.gm > 0 ? p.gm : 4;   // row-blocks per group of the tile walk (option G256_GM; launch_256p)
    const int tpg = GM * nbn;
    const int gidx = bid / tpg;
    const int gfirst = gidx * GM;
    const int gsz = (nbm - gfirst < GM) ? nbm - gfirst : GM;
    const int gin = bid - gidx * tpg;
    const int m0 = (gfirst + gin % gsz) * BM;
    const int n0 = (gin / gsz) * BN;

    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + (size_t)grp * p.strideA;
    const int wgrp = p.wdiv > 1 ? grp / p.wdiv : grp;
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W) + (size_t)wgrp * p.strideW;
    // sparse low part: rows of the whole parameter ahead of this problem's (grouped launches: problem g owns rows g * strideW / 2K ...)
    const int sp_row0 = WS == 3 ? (int)((size_t)wgrp * (size_t)(p.strideW / (2 * (long long)p.K))) + 0 : 0;
    const float* __restrict__ bias = p.bias ? p.bias + (size_t)wgrp * p.strideB : nullptr;
    void* const outp = p.out_table ? p.out_table[grp] : p.out;

    // ---- staging: a wave moves pieces 2 wave, 2 wave + 1 (8 rows x 128 B each) of every half-tile
    const int srow = lane >> 3, pch = lane & 7;
    const T* a_src[2][2];   // [half][piece]
    const T* w_src[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int hr = (wave * 2 + j) * 8 + srow;                    // LDS row inside the half-tile, 0..127
            int gr = m0 + (hr >> 6) * 128 + h * 64 + (hr & 63);
            gr = gr < p.M ? gr : p.M - 1;
            a_src[h][j] = A + (size_t)gr * p.lda + swz(hr, pch) * 8;
            if constexpr (WS == 1) {
                const int col = (hr >> 5) * 64 + h * 32 + (hr & 31);
                w_src[h][j] = W + (size_t)(n0 + col) * (size_t)p.K + swz(hr, pch) * 8;
            } else {
                w_src[h][j] = W + (size_t)(n0 + hr) * (size_t)(p.K * 2) + (size_t)h * p.K + swz(hr, pch) * 8;   // (WS = 3: only h = 0, the hi rows, is used)
            }
        }
    // sparse low part: this wave's KB of kept values (16 rows x 64 B, 64-byte-row swizzle on the source side) and its 256 B of positions, per K-tile
    const char* lo_src = nullptr;
    const char* ix_src = nullptr;
    size_t lo_kstride = 0, ix_kstride = 0;
    if constexpr (WS == 3) {
        const int r = wave * 16 + (lane >> 2);
        lo_src = reinterpret_cast<const char*>(p.Wlo_sp) + ((size_t)(sp_row0 + n0 + r)) * 64 + swz32(r, lane & 3) * 16;
        lo_kstride = (size_t)p.wsp_rows * 64;
        ix_src = reinterpret_cast<const char*>(p.Widx_sp) + ((size_t)((sp_row0 + n0) / 32 + wc) * 64 + lane) * 4;
        ix_kstride = (size_t)(p.wsp_rows / 32) * 256;
    }
    // which: 0 A0, 1 A1, 2 B0, 3 B1 (the half-tile's place in a buffer); kt: K-tile
    auto stage = [&](auto whichc, int kt) {
        constexpr int which = decltype(whichc)::value;
        if constexpr (WS == 3 && which == 3) {
            char* const b1 = smem + (kt & 1) * BUFB + 3 * HALFB;
            glds16(lo_src + (size_t)kt * lo_kstride, b1 + wave * 1024);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ix_src + (size_t)kt * ix_kstride),
                                             (__attribute__((address_space(3))) void*)(b1 + 8192 + wave * 256), 4, 0, 0);
        } else {
            char* const base = smem + (kt & 1) * BUFB + which * HALFB + wave * 2048;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const T* s = which < 2 ? a_src[which & 1][j] : w_src[which & 1][j];
                glds16(s + (size_t)kt * BK, base + j * 1024);
            }
        }
    };
    typedef std::integral_constant<int, 0> HA0;
    typedef std::integral_constant<int, 1> HA1;
    typedef std::integral_constant<int, 2> HB0;
    typedef std::integral_constant<int, 3> HB1;

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int nk = p.K / BK;
    // byte offsets of this lane's fragments inside a buffer: + i * 2048 per row fragment, + HALFB per sub-tile
    unsigned a_lane[2], w_lane[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ra = wr * 64 + fr, rw = wc * 32 + fr;
        a_lane[ks] = (unsigned)(ra * 128 + swz(ra, ks * 4 + fg) * 16);
        w_lane[ks] = 2 * HALFB + (unsigned)(rw * 128 + swz(rw, ks * 4 + fg) * 16);
    }
    f32x4 bpre[NF];   // bias columns, in front of the first DMA (gemm256k_kernel)
    epilogue_bias<EPI, NF>(p, bias, n0 + wc * WN, fg, bpre);
    Fold256In fin;
    float fmu = 0.f, frstd = 0.f;
    if constexpr (FOLDC) fold256_issue(p, m0, n0, tid, fin);   // (fin.s: the tile's 256 s_n, on their way to LDS)
    unsigned lo_lane = 0, ix_lane = 0;   // WS = 3: this lane's 16 bytes of kept values of fragment 0 (fragment 1: + 16 rows x 64 B) and its dword of positions
    if constexpr (WS == 3) {
        const int rw = wc * 32 + fr;
        lo_lane = 3 * HALFB + (unsigned)(rw * 64 + swz32(rw, fg) * 16);
        ix_lane = 3 * HALFB + 8192u + (unsigned)(wave * 256 + lane * 4);
    }
    int lo_idx = 0;

    v8 af[2][4], bf[2][2][2];   // [ks][row fragment of the sub-tile]; [sub-tile g][ks][fragment]
    auto read_a = [&](int h, unsigned bufb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) af[ks][i] = *reinterpret_cast<const v8*>(smem + bufb + h * HALFB + a_lane[ks] + i * 2048);
    };
    auto read_b = [&](auto gc, unsigned bufb) {
        constexpr int g = decltype(gc)::value;
        if constexpr (WS == 3 && g == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[1][0][j] = *reinterpret_cast<const v8*>(smem + bufb + lo_lane + j * 1024);
            lo_idx = *reinterpret_cast<const int*>(smem + bufb + ix_lane);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[g][ks][j] = *reinterpret_cast<const v8*>(smem + bufb + g * HALFB + w_lane[ks] + j * 2048);
        }
    };
    // multiply interval: sub-tile h of the rows against column sub-tile g (plain, 16 MFMAs) ...
    auto mma_q = [&](auto hc, auto gc) {
        constexpr int h = decltype(hc)::value, g = decltype(gc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h * 4 + i][g * 2 + j] = mfma16(bf[g][ks][j], af[ks][i], acc[h * 4 + i][g * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
    // ... or against everything staged of W (32 MFMAs): plain -- both column sub-tiles; split -- hi then lo of the same two fragments in each 32-deep step
    auto mma_h = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (WS == 3) {
            typedef __attribute__((ext_vector_type(16))) _Float16 f16x16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[h * 4 + i][j] = mfma16(bf[0][ks][j], af[ks][i], acc[h * 4 + i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x16 x2k = __builtin_shufflevector(af[0][i], af[1][i], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                acc[h * 4 + i][0] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(bf[1][0][0], x2k, acc[h * 4 + i][0], lo_idx, 0, 0);
                acc[h * 4 + i][1] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(bf[1][0][1], x2k, acc[h * 4 + i][1], lo_idx, 0, 1);
            }
        } else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (WS == 1) acc[h * 4 + i][g * 2 + j] = mfma16(bf[g][ks][j], af[ks][i], acc[h * 4 + i][g * 2 + j]);
                        else acc[h * 4 + i][j] = mfma16(bf[g][ks][j], af[ks][i], acc[h * 4 + i][j]);
                    }
        __builtin_amdgcn_s_setprio(0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
#define M3R_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    // end of a load interval: own reads retired (WAR, and the fragments are there for the MFMAs), then the hand-over barrier
#define M3R_P_LOAD_END() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define M3R_P_MUL_END() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

    // ---- prologue: what the steady state would have issued before K-tile 0
    if constexpr (NPH == 4) {
        stage(HA0{}, 0); stage(HB0{}, 0); stage(HB1{}, 0); stage(HA1{}, 0);
        if (nk > 1) { stage(HA0{}, 1); stage(HB0{}, 1); stage(HB1{}, 1); M3R_VMCNT(8); }
        else M3R_VMCNT(2);
    } else {
        stage(HA0{}, 0); stage(HB0{}, 0); stage(HB1{}, 0); stage(HA1{}, 0);
        if (nk > 1) { stage(HA0{}, 1); stage(HB0{}, 1); M3R_VMCNT(6); }
        else M3R_VMCNT(2);
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (FOLDC) {   // (its loads are older than the DMA the wait above covered)
        fold256_landed(fin);
        fold256_finish(p, m0, n0, tid, fin, fmu, frstd);
        // behind the two buffers: [256] (mean, 1/sigma) | [256] s_n of the tile's columns; every barrier of the K loop lies between these writes and their reads
        char* const fs = smem + 2 * BUFB;
        if (!(tid & 1)) reinterpret_cast<f32x2*>(fs)[tid >> 1] = f32x2{fmu, frstd};
        if (tid < 64) *reinterpret_cast<f32x4*>(fs + 2048 + tid * 16) = fin.s;
    }
    if constexpr (SYNC == 2) {
    if (wr == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind

    // one K-tile; REM = min(2, K-tiles behind this one): which half-tiles are still to be staged, and the counted waits that go with them
    auto ktile = [&](auto remc, int t) {
        constexpr int REM = decltype(remc)::value;
        const unsigned bufb = (t & 1) * BUFB;
        if constexpr (NPH == 4) {
            // phase 0: (a0, b0)
            read_a(0, bufb); read_b(I0{}, bufb);
            if constexpr (REM >= 1) stage(HA1{}, t + 1);
            M3R_P_LOAD_END(); mma_q(I0{}, I0{}); M3R_P_MUL_END();
            // phase 1: (a0, b1); A1 of this K-tile landed
            read_b(I1{}, bufb);
            if constexpr (REM >= 2) stage(HA0{}, t + 2);
            M3R_VMCNT(REM == 2 ? 10 : (REM == 1 ? 8 : 0));
            M3R_P_LOAD_END(); mma_q(I0{}, I1{}); M3R_P_MUL_END();
            // phase 2: (a1, b1)
            read_a(1, bufb);
            if constexpr (REM >= 2) stage(HB0{}, t + 2);
            M3R_P_LOAD_END(); mma_q(I1{}, I1{}); M3R_P_MUL_END();
            // phase 3: (a1, b0) from the registers of phase 0; A0, B0, B1 of the next K-tile landed
            if constexpr (REM >= 2) stage(HB1{}, t + 2);
            if constexpr (REM >= 1) M3R_VMCNT(REM == 2 ? 8 : 2);
            M3R_P_LOAD_END(); mma_q(I1{}, I0{}); M3R_P_MUL_END();
        } else {
            // phase 0: a0 x (b0, b1); A1 of this K-tile landed
            M3R_STAMPP(0);
            read_a(0, bufb); read_b(I0{}, bufb); read_b(I1{}, bufb);
            if constexpr (REM >= 1) { stage(HB1{}, t + 1); stage(HA1{}, t + 1); }
            M3R_STAMPP(1);
            M3R_VMCNT(REM >= 1 ? 8 : 0);
            M3R_STAMPP(2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            M3R_STAMPP(3);
            M3R_P_LOAD_END();
            M3R_STAMPP(4);
            mma_h(I0{});
            M3R_STAMPP(5);
            M3R_P_MUL_END();
            // phase 1: a1 x (b0, b1); A0, B0, B1 of the next K-tile landed
            M3R_STAMPP(6);
            read_a(1, bufb);
            if constexpr (REM >= 2) { stage(HA0{}, t + 2); stage(HB0{}, t + 2); }
            M3R_STAMPP(7);
            if constexpr (REM >= 1) M3R_VMCNT(REM == 2 ? 6 : 2);
            M3R_STAMPP(8);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            M3R_STAMPP(9);
            M3R_P_LOAD_END();
            M3R_STAMPP(10);
            mma_h(I1{});
            M3R_STAMPP(11);
            M3R_P_MUL_END();
            M3R_STAMPP(12);
        }
    };
    int t = 0;
    for (; t + 2 < nk; ++t) ktile(std::integral_constant<int, 2>{}, t);
    if (t + 1 < nk) { ktile(std::integral_constant<int, 1>{}, t); ++t; }
    ktile(std::integral_constant<int, 0>{}, t);
    if (wr == 0) __builtin_amdgcn_s_barrier();
    } else {
        // load blocks of the two phases of K-tile t (the same for both groups; the schedule and the counted waits of the SYNC = 2 form)
        auto load0 = [&](auto remc, int t) {
            constexpr int REM = decltype(remc)::value;
            const unsigned bufb = (t & 1) * BUFB;
            read_a(0, bufb); read_b(I0{}, bufb); read_b(I1{}, bufb);
            if constexpr (REM >= 1) { stage(HB1{}, t + 1); stage(HA1{}, t + 1); }
            M3R_VMCNT(REM >= 1 ? 8 : 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto load1 = [&](auto remc, int t) {
            constexpr int REM = decltype(remc)::value;
            const unsigned bufb = (t & 1) * BUFB;
            read_a(1, bufb);
            if constexpr (REM >= 2) { stage(HA0{}, t + 2); stage(HB0{}, t + 2); }
            if constexpr (REM >= 1) M3R_VMCNT(REM == 2 ? 6 : 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        if (wr == 0) {
            auto ktile = [&](auto remc, int t) {
                load0(remc, t); mma_h(I0{}); M3R_P_MUL_END();
                load1(remc, t); mma_h(I1{}); M3R_P_MUL_END();
            };
            int t = 0;
            for (; t + 2 < nk; ++t) ktile(std::integral_constant<int, 2>{}, t);
            if (t + 1 < nk) { ktile(std::integral_constant<int, 1>{}, t); ++t; }
            ktile(std::integral_constant<int, 0>{}, t);
        } else {
            auto ktile = [&](auto remc, int t) {
                if (t > 0) { mma_h(I1{}); __builtin_amdgcn_sched_barrier(0); }   // the second half of K-tile t - 1, on the fragments read in the last interval
                load0(remc, t); M3R_P_MUL_END();
                mma_h(I0{}); __builtin_amdgcn_sched_barrier(0);
                load1(remc, t); M3R_P_MUL_END();
            };
            int t = 0;
            for (; t + 2 < nk; ++t) ktile(std::integral_constant<int, 2>{}, t);
            if (t + 1 < nk) { ktile(std::integral_constant<int, 1>{}, t); ++t; }
            ktile(std::integral_constant<int, 0>{}, t);
            mma_h(I1{});
        }
    }
#undef M3R_VMCNT
#undef M3R_P_LOAD_END
#undef M3R_P_MUL_END

    if constexpr (FOLDC) fold256_apply_staged<NF, MF>(smem + 2 * BUFB, wr * 128 + fr, wc * WN + fg * 4, acc);
    epilogue_tile<T, EPI, NF, MF, 4, false, NoLnFold, FOLDP>(p, outp, bias, m0 + wr * 128 + fr, n0 + wc * WN, fg, acc, NoLnFold{}, bpre);
    if constexpr (PERS == 0) break;
    }
}

// blocks of a persistent launch: one per CU (the kernels hold a CU's whole LDS / half its registers: one block is resident per CU)
static int persistent_blocks() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n;
}
// PERSIST (A/B instrument): 1 the chip-filling kernels walk their tiles in a persistent loop when the launch has more than one round, 0 (default) never.
// Measured (profiles/r06_gemm_persist_ab.txt, one process, interleaved rounds, identical bits): x0.96 ... x1.04 per shape, x1.003 over the 16 shapes; step 526.1 (0) vs 523.6 (1) views/s.
static int persist_mode() { return opt(OPT_PERSIST); }

template <class T, int EPI, int WS, int BN, int NPH, int SYNC = 2>
static int launch_256p(const GemmArgs& a_in, hipStream_t s) {
    GemmArgs a = a_in;
    a.gm = opt(OPT_G256_GM);
    const int nbn = a.N / BN, nbm = (a.M + 255) / 256;
    const size_t lds = (size_t)2 * 4 * 128 * 64 * sizeof(T);
    const long nwork = (long)nbm * nbn * (a.batch > 1 ? a.batch : 1);
    if (persist_mode() != 0 && !a.fold256 && nwork > persistent_blocks() && nwork < (1l << 30)) {
        static bool attr_set_p = false;
        if (!attr_set_p) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<T, EPI, WS, BN, NPH, SYNC, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
            attr_set_p = true;
        }
        hipLaunchKernelGGL((gemm256p_kernel<T, EPI, WS, BN, NPH, SYNC, 1>), dim3(persistent_blocks()), dim3(512), lds, s, a);
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
    if constexpr (WS == 1 && BN == 256 && NPH == 2 && SYNC == 2 && sizeof(T) == 2 && (EPI == EPI_STORE16_GELU || EPI == EPI_RESID_F32)) {
        if (a.fold256) {   // r06: the LN-fold forms (fc1 consumer, fc2 producer of the fp16wa mode)
            static bool attr_set_f = false;
            if (!attr_set_f) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<T, EPI, WS, BN, NPH, SYNC, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 4096) != hipSuccess) return 1;
                attr_set_f = true;
            }
            hipLaunchKernelGGL((gemm256p_kernel<T, EPI, WS, BN, NPH, SYNC, 0, 1>), dim3(nbm * nbn, a.batch > 1 ? a.batch : 1), dim3(512), lds + 4096, s, a);   // + (mean, 1/sigma), s_n
            return hipGetLastError() == hipSuccess ? 0 : 1;
        }
    }
    if (a.fold256) return 1;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<T, EPI, WS, BN, NPH, SYNC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm256p_kernel<T, EPI, WS, BN, NPH, SYNC>), dim3(nbm * nbn, a.batch > 1 ? a.batch : 1), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

#ifdef M3R_GEMM_LAB   // gemm256n_kernel: one phase per K-tile (measured equal; scripts/probes/kloop_lab.hip)
#include "../../scripts/probes/lab/gemm256n.inc"
#endif

// ------------------------------------------------------------------------------------------------------------------
// gemm256s_kernel (r05): split weights with a 2:4-sparse low part on 256 x 256 tiles -- gemm256p_kernel's two-phase K loop with FIVE half-tiles per K-tile.
//   Why another tile: with the sparse low part a 256 x 128 tile (gemm256p_kernel<.., WS = 3>) issues 24 matrix instructions per phase (~410 cycles) against a
//   load interval of ~500-600 (its 15 fragment reads, 4 DMA pieces and their waits): the K-tile takes the ~2600 cycles of the dense form and only 1.7 % of the
//   step came back (profiles/r05_sparse_lo_ab.txt).  256 columns double the matrix work per staged byte: 48 instructions per phase (32 dense hi + 16 sparse lo).
//   LDS per K-tile: A 32 KB + W_hi 32 KB + kept W_lo values 16 KB = 80 KB, two buffers = ALL 160 KB of the CU -- the 2-bit positions (2 dwords per lane and
//   K-tile) therefore come straight from L2 into registers (inline-asm loads on the same in-order vmcnt queue as the DMA pieces, issued a K-tile ahead).
//   Half-tiles: A0 A1 (rows, as gemm256p) | H0 H1 (hi weight rows of every wave's column sub-tile g) | LO (the kept low values of all 256 weight rows, 64 B each).
//   Schedule per K-tile t (lag-1 restaging as gemm256p; every wave issues 2 pieces per half-tile):
//     phase 0: read a0, h0, h1, lo | stage H1, LO, A1 of t+1 | vmcnt(6): A1(t) and the positions of t landed | 48 matrix instructions
//     phase 1: read a1            | stage A0, H0 of t+2, load the positions of t+1 | vmcnt(8): H1, LO, A0, H0 of t+1 landed | 48 matrix instructions
// Numerics: those of gemm256p_kernel<.., WS = 3> (hi products in k order, then the sparse low product of the K-tile): identical bits between the two.
// PERS = 1 (r06): persistent tiles, as gemm256p_kernel (the positions' first load of the next tile goes out behind this tile's stores; in-order vmcnt covers it).
template <class T, int EPI, int PERS = 0, int FOLD = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) gemm256s_kernel(const GemmArgs p) {
    static_assert(sizeof(T) == 2 && !__is_same(T, bf16_t), "fp16 only");
    constexpr bool FOLDC = FOLD != 0 && (EPI == EPI_STORE16 || EPI == EPI_STORE16_GELU || EPI == EPI_QKV_ROPE);   // r06: LN fold, as gemm256p_kernel
    constexpr bool FOLDP = FOLD != 0 && EPI == EPI_RESID_F32;
    static_assert(FOLD == 0 || ((FOLDC || FOLDP) && PERS == 0), "fold256: one tile per block");
    typedef typename Vec<T>::v8 v8;
    typedef __attribute__((ext_vector_type(16))) _Float16 f16x16;
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int WN = 64, MF = 8, NF = 4;
    constexpr unsigned HALFB = 128 * BK * 2;               // 16 KB
    constexpr unsigned BUFB = 5 * HALFB;                   // A0 A1 H0 H1 LO
    extern __shared__ __attribute__((aligned(16))) char smem[];

    for (int work = blockIdx.x;; work += gridDim.x) {   // PERS = 0: one pass (the body is not re-indented)
    int tid_ = threadIdx.x;
    // PERS: everything derived from the lane is recomputed per tile (a few integer instructions) -- hoisted out of the tile loop it would have to live across the
    // epilogue, and the kernel sits at the 256-register cap of two waves per SIMD (spills)
    if constexpr (PERS != 0) asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nbn = p.N / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;
    const int nwork = PERS ? nwg * (p.batch > 1 ? p.batch : 1) : 0;
    if constexpr (PERS != 0) { if (work >= nwork) break; }
    const int grp = PERS ? work / nwg : (int)blockIdx.y;
    int bid = PERS ? work - grp * nwg : (int)blockIdx.x;
    {
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int GM = p.gm > 0 ? p.gm : 4;   // (option G256_GM; launch_256s)
    const int tpg = GM * nbn;
    const int gidx = bid / tpg;
    const int gfirst = gidx * GM;
    const int gsz = (nbm - gfirst < GM) ? nbm - gfirst : GM;
    const int gin = bid - gidx * tpg;
    const int m0 = (gfirst + gin % gsz) * BM;
    const int n0 = (gin / gsz) * BN;

    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + (size_t)grp * p.strideA;
    const int wgrp = p.wdiv > 1 ? grp / p.wdiv : grp;
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W) + (size_t)wgrp * p.strideW;
    const float* __restrict__ bias = p.bias ? p.bias + (size_t)wgrp * p.strideB : nullptr;
    void* const outp = p.out_table ? p.out_table[grp] : p.out;
    const int sp_row0 = (int)((size_t)wgrp * (size_t)(p.strideW / (2 * (long long)p.K)));

    // ---- staging sources: pieces 2 wave, 2 wave + 1 of every half-tile
    const int srow = lane >> 3, pch = lane & 7;
    const T* a_src[2][2];
    const T* h_src[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int hr = (wave * 2 + j) * 8 + srow;
            int gr = m0 + (hr >> 6) * 128 + h * 64 + (hr & 63);
            gr = gr < p.M ? gr : p.M - 1;
            a_src[h][j] = A + (size_t)gr * p.lda + swz(hr, pch) * 8;
            const int col = (hr >> 5) * 64 + h * 32 + (hr & 31);
            h_src[h][j] = W + (size_t)(n0 + col) * (size_t)(p.K * 2) + swz(hr, pch) * 8;
        }
    const char* lo_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave * 2 + j) * 16 + (lane >> 2);
        lo_src[j] = reinterpret_cast<const char*>(p.Wlo_sp) + ((size_t)(sp_row0 + n0 + r)) * 64 + swz32(r, lane & 3) * 16;
    }
    const size_t lo_kstride = (size_t)p.wsp_rows * 64;
    const size_t ix_kstride = (size_t)(p.wsp_rows / 32) * 256;
    const char* const ix_base = reinterpret_cast<const char*>(p.Widx_sp) + ((size_t)((sp_row0 + n0) / 32 + 2 * wc)) * 256;   // this wave's two 32-row blocks
    const unsigned ix_lane = (unsigned)(lane * 4);

    // which: 0 A0, 1 A1, 2 H0, 3 H1, 4 LO
    auto stage = [&](auto whichc, int kt) {
        constexpr int which = decltype(whichc)::value;
        char* const base = smem + (kt & 1) * BUFB + which * HALFB + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (which == 4) glds16(lo_src[j] + (size_t)kt * lo_kstride, base + j * 1024);
            else glds16((which < 2 ? a_src[which & 1][j] : h_src[which & 1][j]) + (size_t)kt * BK, base + j * 1024);
        }
    };
    typedef std::integral_constant<int, 0> HA0;
    typedef std::integral_constant<int, 1> HA1;
    typedef std::integral_constant<int, 2> HH0;
    typedef std::integral_constant<int, 3> HH1;
    typedef std::integral_constant<int, 4> HLO;

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int nk = p.K / BK;
    unsigned a_lane[2], w_lane[2], lo_lane;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ra = wr * 64 + fr, rw = wc * 32 + fr;
        a_lane[ks] = (unsigned)(ra * 128 + swz(ra, ks * 4 + fg) * 16);
        w_lane[ks] = 2 * HALFB + (unsigned)(rw * 128 + swz(rw, ks * 4 + fg) * 16);
    }
    {
        const int rl = wc * 64 + fr;
        lo_lane = 4 * HALFB + (unsigned)(rl * 64 + swz32(rl, fg) * 16);
    }
    Fold256In fin;
    float fmu = 0.f, frstd = 0.f;
    if constexpr (FOLDC) fold256_issue(p, m0, n0, tid, fin);
    v8 af[2][4], bh[2][2][2], bl[4];
    int ix_nxt[2], ix_cur[2] = {0, 0};
    auto read_a = [&](int h, unsigned bufb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) af[ks][i] = *reinterpret_cast<const v8*>(smem + bufb + h * HALFB + a_lane[ks] + i * 2048);
    };
    auto read_w = [&](unsigned bufb) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) bh[g][ks][j] = *reinterpret_cast<const v8*>(smem + bufb + g * HALFB + w_lane[ks] + j * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) bl[j] = *reinterpret_cast<const v8*>(smem + bufb + lo_lane + j * 1024);
    };
    // the positions of K-tile kt: two dwords per lane, straight from L2 (scalar base + 32-bit lane offset); their registers are unprotected until the counted
    // wait that covers them (the volatile statements below keep their order: load ... wait ... move)
    auto load_ix = [&](int kt) {
        const char* const b0 = ix_base + (size_t)kt * ix_kstride;
        asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:256" : "=&v"(ix_nxt[0]), "=&v"(ix_nxt[1]) : "v"(ix_lane), "s"(b0) : "memory");
    };
    auto take_ix = [&]() {
        // (%0 is written before %3 is read: early-clobber, or the allocator may give ix_cur[0] the register of ix_nxt[1] -- it did in <EPI_F32, PERS = 1>)
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(ix_cur[0]), "=v"(ix_cur[1]) : "v"(ix_nxt[0]), "v"(ix_nxt[1]));
    };
    auto mma_h = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[h * 4 + i][g * 2 + j] = mfma16(bh[g][ks][j], af[ks][i], acc[h * 4 + i][g * 2 + j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f16x16 x2k = __builtin_shufflevector(af[0][i], af[1][i], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
            acc[h * 4 + i][0] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(bl[0], x2k, acc[h * 4 + i][0], ix_cur[0], 0, 0);
            acc[h * 4 + i][1] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(bl[1], x2k, acc[h * 4 + i][1], ix_cur[0], 0, 1);
            acc[h * 4 + i][2] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(bl[2], x2k, acc[h * 4 + i][2], ix_cur[1], 0, 0);
            acc[h * 4 + i][3] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(bl[3], x2k, acc[h * 4 + i][3], ix_cur[1], 0, 1);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
#define M3R_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define M3R_P_LOAD_END() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define M3R_P_MUL_END() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

    // ---- prologue: what the steady state would have issued before K-tile 0 (in its order)
    stage(HA0{}, 0); stage(HH0{}, 0); stage(HH1{}, 0); stage(HLO{}, 0); stage(HA1{}, 0);
    if (nk > 1) { stage(HA0{}, 1); stage(HH0{}, 1); load_ix(0); M3R_VMCNT(8); }
    else { load_ix(0); M3R_VMCNT(4); }
    __builtin_amdgcn_s_barrier();
    if constexpr (FOLDC) {   // (its loads are older than the DMA the wait above covered)
        fold256_landed(fin);
        fold256_finish(p, m0, n0, tid, fin, fmu, frstd);
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind

    auto ktile = [&](auto remc, int t) {
        constexpr int REM = decltype(remc)::value;   // min(2, K-tiles behind this one)
        const unsigned bufb = (t & 1) * BUFB;
        // phase 0
        read_a(0, bufb); read_w(bufb);
        if constexpr (REM >= 1) { stage(HH1{}, t + 1); stage(HLO{}, t + 1); stage(HA1{}, t + 1); }
        M3R_VMCNT(REM >= 1 ? 6 : 0);
        take_ix();
        M3R_P_LOAD_END(); mma_h(I0{}); M3R_P_MUL_END();
        // phase 1
        read_a(1, bufb);
        if constexpr (REM >= 2) { stage(HA0{}, t + 2); stage(HH0{}, t + 2); }
        if constexpr (REM >= 1) { load_ix(t + 1); M3R_VMCNT(REM == 2 ? 8 : 4); }
        M3R_P_LOAD_END(); mma_h(I1{}); M3R_P_MUL_END();
    };
    int t = 0;
    for (; t + 2 < nk; ++t) ktile(std::integral_constant<int, 2>{}, t);
    if (t + 1 < nk) { ktile(std::integral_constant<int, 1>{}, t); ++t; }
    ktile(std::integral_constant<int, 0>{}, t);
    if (wr == 0) __builtin_amdgcn_s_barrier();
#undef M3R_VMCNT
#undef M3R_P_LOAD_END
#undef M3R_P_MUL_END

    if constexpr (FOLDC) {
        f32x4 fb[NF];
        fold256_apply<NF, MF>(p, bias, smem, tid, wr * 128 + fr, n0 + wc * WN + fg * 4, fmu, frstd, acc, fb);
        epilogue_tile<T, EPI, NF, MF, 4, false, NoLnFold, false>(p, outp, bias, m0 + wr * 128 + fr, n0 + wc * WN, fg, acc, NoLnFold{}, fb);
    } else
    epilogue_tile<T, EPI, NF, MF, 4, false, NoLnFold, FOLDP>(p, outp, bias, m0 + wr * 128 + fr, n0 + wc * WN, fg, acc, NoLnFold{});
    if constexpr (PERS == 0) break;
    }
}

template <class T, int EPI>
static int launch_256s(const GemmArgs& a_in, hipStream_t s) {
    GemmArgs a = a_in;
    a.gm = opt(OPT_G256_GM);
    const int nbn = a.N / 256, nbm = (a.M + 255) / 256;
    const size_t lds = (size_t)2 * 5 * 128 * 64 * sizeof(T);   // 163840 B: the CU's whole LDS
    const long nwork = (long)nbm * nbn * (a.batch > 1 ? a.batch : 1);
    if (persist_mode() != 0 && !a.fold256 && nwork > persistent_blocks() && nwork < (1l << 30)) {
        static bool attr_set_p = false;
        if (!attr_set_p) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_kernel<T, EPI, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
            attr_set_p = true;
        }
        hipLaunchKernelGGL((gemm256s_kernel<T, EPI, 1>), dim3(persistent_blocks()), dim3(512), lds, s, a);
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
    if constexpr (EPI == EPI_STORE16 || EPI == EPI_QKV_ROPE || EPI == EPI_RESID_F32) {
        if (a.fold256) {   // r06: the LN-fold forms (qkv / projq consumers, proj producer)
            static bool attr_set_f = false;
            if (!attr_set_f) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_kernel<T, EPI, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
                attr_set_f = true;
            }
            hipLaunchKernelGGL((gemm256s_kernel<T, EPI, 0, 1>), dim3(nbm * nbn, a.batch > 1 ? a.batch : 1), dim3(512), lds, s, a);
            return hipGetLastError() == hipSuccess ? 0 : 1;
        }
    }
    if (a.fold256) return 1;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_kernel<T, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm256s_kernel<T, EPI>), dim3(nbm * nbn, a.batch > 1 ? a.batch : 1), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

#ifdef M3R_GEMM_LAB   // gemm256pp_kernel / gemm256q_kernel / gemm256w_kernel: persistent tiles, one barrier per K-tile, four-wave 128 x 128 tiles (measured, not used)
#include "../../scripts/probes/lab/gemm256_pp_q_w.inc"
#endif

// ------------------------------------------------------------------------------------------------------------------
// M = 768 launches with N = 768 (proj, fc2, projq of the memory update): 48 x 48 tiles = exactly 256 blocks, ONE per CU,
// 9 waves (3 x 3, wave tile 16 x 16).  A cycle trace of the 64 x 64 kernels (scripts/probes/gemm_trace.hip) shows where a
// K-tile's 0.35 us go: every wave spends ~250 cycles ISSUING its three 1 KB LDS-DMA instructions (~80 cycles each), ~270
// on its ten ds_read_b128 and only ~130 on its eight MFMAs -- per-wave instruction issue, in lock step, not bandwidth.
// With 144 tiles of 64 x 64 the other 112 CUs idle; 48 x 48 tiles put all 256 CUs to work with 2 DMA + 6 LDS reads + 4
// MFMAs per wave and K-tile.  Measured (split weights, M = N = 768): proj 8.0 -> 6.9 us, fc2 20.6 -> 18.9 us, same bits.
// (The K-loop slope stays ~0.33 us per tile even here, so per-wave issue is not the whole story either; the gain is in
// the fixed part.)  Plain weights (r03): 12 pieces over 9 waves -- waves 0-2 issue two per tile, the others one, each counting its own.
// r04: K-tile depth BK = 128 where K allows (every launch of the scene: K = 768 / 3072).  The K loop of these launches is a latency chain per
// tile -- counted wait, block barrier, DMA issue, fragment reads, two dependent MFMAs per weight part -- of ~0.33 us whatever the tile holds
// (profiles/r02_gemm_small_*.txt: the slope is per TILE, not per byte); 128-deep tiles walk the chain half as often.  256-byte LDS rows, 4 rows per
// DMA piece, swz128; the fragments of the four 32-deep steps are read up front, the accumulation order (k ascending, hi before lo per step) and
// therefore the bits are those of every other tile shape.
template <class T, int EPI, int WS, int NST, int BK>
__global__ void __launch_bounds__(576) gemm48_kernel(const GemmArgs p) {
    typedef typename Vec<T>::v8 v8;
    static_assert(BK == 64 || BK == 128, "K-tile depth");
    constexpr int BM = 48, BN = 48, NW = 9, CPR = BK / 8, RPP = 64 / CPR, KS = BK / 32;
    constexpr int ROWS = BM + WS * BN;                 // staging region: A rows, then W rows (hi, then lo)
    constexpr int NPIECE = ROWS / RPP;                 // 18 (split: 2 per wave) or 12 (plain: waves 0-2 issue 2, the others 1)
    constexpr int STAGE = ROWS * BK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* const lds = reinterpret_cast<T*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 3, wn = wave - wm * 3;
    const int nbn = p.N / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m0 = (bid % nbm) * BM;                    // row-block fastest: the blocks of an XCD share weight panels
    const int n0 = (bid / nbm) * BN;
    const int grp = blockIdx.y;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + (size_t)grp * p.strideA;
    const int wgrp = p.wdiv > 1 ? grp / p.wdiv : grp;
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W) + (size_t)wgrp * p.strideW;
    const float* __restrict__ bias = p.bias ? p.bias + (size_t)wgrp * p.strideB : nullptr;
    void* const outp = p.out_table ? p.out_table[grp] : p.out;

    // pieces dealt round-robin: piece = t * NW + wave (split weights: 18 pieces, 2 per wave)
    constexpr int TMAX = (NPIECE + NW - 1) / NW;
    constexpr int NFULL = NPIECE - NW * (TMAX - 1);   // waves that issue TMAX pieces per tile; the others TMAX - 1 (their counted vmcnt differs)
    const bool full_wave = wave < NFULL;
    const int srow = lane / CPR, pch = lane % CPR;
    const T* src[TMAX];
    bool has[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const int piece = t * NW + wave;
        has[t] = piece < NPIECE;
        const int r = (has[t] ? piece : 0) * RPP + srow;
        if (r < BM) {
            int gr = m0 + r;
            gr = gr < p.M ? gr : p.M - 1;
            src[t] = A + (size_t)gr * p.lda + swzk<BK>(r, pch) * 8;
        } else {
            const int rr = r - BM, part = rr / BN, wrow = rr - part * BN;
            src[t] = W + (size_t)(n0 + wrow) * (size_t)(p.K * WS) + (size_t)part * p.K + swzk<BK>(rr, pch) * 8;
        }
    }
    auto stage = [&](int kt, int buf) {
        T* base = lds + buf * STAGE;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (has[t]) glds16(src[t] + kt * BK, base + (t * NW + wave) * RPP * BK);
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = p.K / BK;
    EpiPre<1> pre;
    epilogue_prefetch<T, EPI, 1>(p, outp, bias, m0 + wm * 16 + fr, n0 + wn * 16, fg, pre);
    constexpr bool LNF = EPI == EPI_STORE16 || EPI == EPI_STORE16_GELU;
    float* const sm_ln = reinterpret_cast<float*>(smem + (size_t)NST * STAGE * sizeof(T));
    int a_off[KS], w_off[KS][WS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int r = wm * 16 + fr;
        a_off[ks] = r * BK + swzk<BK>(r, ks * 4 + fg) * 8;
#pragma unroll
        for (int part = 0; part < WS; ++part) {
            const int rr = part * BN + wn * 16 + fr;
            w_off[ks][part] = (BM + rr) * BK + swzk<BK>(rr, ks * 4 + fg) * 8;
        }
    }
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) stage(t, t);
    if constexpr (LNF) {   // LN fold: the row statistics are gathered while the ring prologue is in flight
        if (p.ln_stats != nullptr) ln_fold_rows<BM, 64 * NW>(p, m0, sm_ln, tid, n0 == 0);
    }
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt landed: at most NST-2 younger tiles x TMAX loads may still be in flight (tail: drain)
        if (kt + NST - 2 < nk) {
            if (NFULL == NW || full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * TMAX) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (TMAX - 1)) : "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nt = kt + NST - 1;
        if (nt < nk) {
            int nb = buf + NST - 1;
            nb = nb >= NST ? nb - NST : nb;
            stage(nt, nb);
        }
        const T* base = lds + buf * STAGE;
        v8 af[KS], wf[KS][WS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            af[ks] = *reinterpret_cast<const v8*>(base + a_off[ks]);
#pragma unroll
            for (int part = 0; part < WS; ++part) wf[ks][part] = *reinterpret_cast<const v8*>(base + w_off[ks][part]);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int part = 0; part < WS; ++part) acc = mfma16(wf[ks][part], af[ks], acc);
        buf = buf + 1 == NST ? 0 : buf + 1;
    }
    const int m = m0 + wm * 16 + fr;
    if (m < p.M) {
        f32x4 v[1] = {acc};
        float mu = 0.f, rstd = 0.f;
        if constexpr (LNF) {
            if (p.ln_stats != nullptr) {
                mu = sm_ln[BM * 12 * 2 + (wm * 16 + fr) * 2];
                rstd = sm_ln[BM * 12 * 2 + (wm * 16 + fr) * 2 + 1];
            }
        }
        epilogue_row<T, EPI, 1>(p, outp, bias, m, n0 + wn * 16, fg, v, &pre, mu, rstd);
    }
}

static thread_local int g_gemm48_bk = 64;   // K-tile depth of this thread's last gemm48 launch: the two depths are two symbols of a kernel trace (pick_name48)
template <class T, int EPI, int WS, int BK>
static int launch_48k(const GemmArgs& a, hipStream_t s) {
    constexpr int NST = BK == 128 ? (WS == 2 ? 4 : 6) : 6;   // 4 x 36 KB (split) / 6 x 24 KB (plain) of 128-deep stages; 6 x 18 / 12 KB at 64
    const int nbn = a.N / 48, nbm = (a.M + 47) / 48;
    const size_t lds = (size_t)NST * (48 + WS * 48) * BK * sizeof(T) + ln_fold_lds_bytes<48, 576>();
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm48_kernel<T, EPI, WS, NST, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            fprintf(stderr, "must3r_hip: gemm48 (K-tile %d): %zu bytes of dynamic LDS refused\n", BK, lds);
            return 1;
        }
        attr_set = true;
    }
    g_gemm48_bk = BK;
    hipLaunchKernelGGL((gemm48_kernel<T, EPI, WS, NST, BK>), dim3(nbm * nbn, a.batch > 1 ? a.batch : 1), dim3(576), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
// M3R_BK128 (A/B instrument): 0 = 64-deep K-tiles everywhere (the r03 kernels).  Measured (profiles/r04_bk128_m768.txt): same
// bits; fc2 (K = 3072) 17.4 -> 14.1 us plain, 19.0 -> 17.9 us split; the K = 768 launches +-0.2 us (their 12 tiles are not what they spend their
// time on); one-scene-at-a-time 331 -> 336 views/s.  The same depth in the 64 x 64 ring kernels (2 x 128-deep stages): qkv 11.1 -> 15.1 us,
// K|V 10.2 -> 14.4 us, fc1 13.3 -> 12.5 us, scene 326 views/s -- not kept.
static int bk128_mode() { return opt(OPT_BK128); }
template <class T, int EPI, int WS>
static int launch_48(const GemmArgs& a, hipStream_t s) {
    if (a.K % 128 == 0 && a.K >= 256 && bk128_mode() != 0) return launch_48k<T, EPI, WS, 128>(a, s);
    return launch_48k<T, EPI, WS, 64>(a, s);
}

// ------------------------------------------------------------------------------------------------------------------
// M = 768 launches with N >= 2304 (qkv, fc1 of the memory update; feedback fc1): 96 x 96 tiles, ONE block per CU, 9 waves
// (3 x 3, wave tile 32 x 32), split weights.  Why this shape: these launches are bound by the CU's L2 -> LDS path (one 1 KB
// LDS-DMA instruction costs its wave ~60-80 issue cycles = ~64 B/clk/CU) and by the tail of the grid, not by MFMA rate.
//   64 x 64 tiles: 24 KB staged per 4096 MACs/k, 576 (fc1) / 432 (qkv) tiles on 512 block slots -> 2 rounds / uneven CUs
//   96 x 96 tiles: 36 KB staged per 9216 MACs/k (1.5 x the bytes for 2.25 x the work), 768 x 3072 = exactly 256 tiles
// Per K-tile a wave issues 4 DMA pieces, 12 ds_read_b128 and 16 MFMAs; 4-slot ring (144 KB), counted vmcnt, one barrier
// per tile.  The fragment reads of tile kt are issued BEFORE the DMA of tile kt+3 so that they are in flight while the
// DMA instructions issue.  Accumulation order per output (k ascending, hi before lo per 32-deep step) as everywhere else:
// same bits as the other tile shapes.
// (A split-K form -- gridDim.z K ranges, fp32 partial slabs, the residual update folded into the next LayerNorm -- was built in r02 for
// the K = 3072 fc2 of the one-view update and removed in r03: in the scene it was a wash, the slabs cost the LayerNorm what the GEMM saved.)
template <class T, int EPI, int NST, int PF>
__global__ void __launch_bounds__(576) gemm96_kernel(const GemmArgs p) {
    typedef typename Vec<T>::v8 v8;
    constexpr int BM = 96, BN = 96, BK = 64, NW = 9, RPP = 8, WS = 2;
    constexpr int ROWS = BM + WS * BN;                 // 288 rows per stage: A, W_hi, W_lo
    constexpr int TP = ROWS / RPP / NW;                // 4 DMA pieces per wave and K-tile
    static_assert(TP * NW * RPP == ROWS, "every wave issues the same number of DMA pieces (counted vmcnt)");
    constexpr int STAGE = ROWS * BK;
    constexpr int MF = 2, NF = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* const lds = reinterpret_cast<T*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 3, wn = wave - wm * 3;
    const int nbn = p.N / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m0 = (bid % nbm) * BM;                    // row-block fastest: the blocks of an XCD share weight panels
    const int n0 = (bid / nbm) * BN;
    const int grp = blockIdx.y;
    const int nk = p.K / BK;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + (size_t)grp * p.strideA;
    const int wgrp = p.wdiv > 1 ? grp / p.wdiv : grp;
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W) + (size_t)wgrp * p.strideW;
    const float* __restrict__ bias = p.bias ? p.bias + (size_t)wgrp * p.strideB : nullptr;
    void* const outp = p.out_table ? p.out_table[grp] : p.out;

    // pieces dealt round-robin: piece = t * NW + wave; rows [0,96) = A, [96,192) = W_hi, [192,288) = W_lo
    const int srow = lane >> 3, pch = lane & 7;
    const T* src[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
        const int r = (t * NW + wave) * RPP + srow;
        if (r < BM) {
            int gr = m0 + r;
            gr = gr < p.M ? gr : p.M - 1;
            src[t] = A + (size_t)gr * p.lda + swz(r, pch) * 8;
        } else {
            const int rr = r - BM, part = rr / BN, wrow = rr - part * BN;
            src[t] = W + (size_t)(n0 + wrow) * (size_t)(p.K * WS) + (size_t)part * p.K + swz(rr, pch) * 8;
        }
    }
    auto stage = [&](int kt, int buf) {
        T* base = lds + buf * STAGE;
#pragma unroll
        for (int t = 0; t < TP; ++t) glds16(src[t] + kt * BK, base + (t * NW + wave) * RPP * BK);
    };
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    EpiPre<NF> pre[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) epilogue_prefetch<T, EPI, NF>(p, outp, bias, m0 + wm * 32 + i * 16 + fr, n0 + wn * 32, fg, pre[i]);
    constexpr bool LNF = EPI == EPI_STORE16 || EPI == EPI_STORE16_GELU || EPI == EPI_QKV_ROPE;
    float* const sm_ln = reinterpret_cast<float*>(smem + (size_t)NST * STAGE * sizeof(T));
    int a_off[2][MF], w_off[2][WS][NF];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int r = wm * 32 + i * 16 + fr;
            a_off[ks][i] = r * BK + swz(r, ks * 4 + fg) * 8;
        }
#pragma unroll
        for (int part = 0; part < WS; ++part)
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int rr = part * BN + wn * 32 + j * 16 + fr;
                w_off[ks][part][j] = (BM + rr) * BK + swz(rr, ks * 4 + fg) * 8;
            }
    }
    if constexpr (PF) {
        // Software-pipelined form.  In the loop below every wave does, per K-tile, [4 DMA][12 LDS reads][16 MFMAs] one after the other
        // and all nine waves do it in lock step behind the barrier: the DMA burst alone keeps the CU's L2->LDS path busy for ~600
        // cycles (36 KB at ~64 B/clk) during which no MFMA issues.  Here the three instruction streams are interleaved inside each
        // wave: the MFMAs of tile kt (fragments read one iteration earlier) are issued in four groups of four, and after each group
        // one DMA piece of tile kt+NST and three fragment reads of tile kt+1 -- the matrix pipe works through a group while the
        // wave sits in the memory instructions.  All NST slots are prefetch depth (tile kt's slot is refilled while tile kt is
        // multiplied from registers).  Same accumulation order (k-slot 0 hi, 0 lo, 1 hi, 1 lo) -> same bits.
        v8 fa[2][2][MF], fw[2][2][WS][NF];   // [set][ks]
        auto read_group = [&](auto setc, int buf, auto gc) {   // 3 of the 12 fragment reads of a tile: group g = (ks, part)
            constexpr int set = decltype(setc)::value, g = decltype(gc)::value, ks = g >> 1, part = g & 1;
            const T* base = lds + buf * STAGE;
#pragma unroll
            for (int j = 0; j < NF; ++j) fw[set][ks][part][j] = *reinterpret_cast<const v8*>(base + w_off[ks][part][j]);
            fa[set][ks][part] = *reinterpret_cast<const v8*>(base + a_off[ks][part]);   // MF == WS == 2: one activation fragment per group
        };
        auto mma_group = [&](auto setc, auto gc) {
            constexpr int set = decltype(setc)::value, g = decltype(gc)::value, ks = g >> 1, part = g & 1;
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = mfma16(fw[set][ks][part][j], fa[set][ks][i], acc[i][j]);
        };
        static_assert(MF == 2 && WS == 2 && TP == 4, "group split");
#pragma unroll
        for (int t = 0; t < NST; ++t)
            if (t < nk) stage(t, t);
        if constexpr (LNF) {   // LN fold: the row statistics are gathered while the ring prologue is in flight
            if (p.ln_stats != nullptr) ln_fold_rows<BM, 64 * NW>(p, m0, sm_ln, tid, n0 == 0);
        }
        // tile 0 landed: up to NST-1 younger tiles in flight
        if (nk >= NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * TP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_group(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{});
        read_group(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 1>{});
        read_group(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 2>{});
        read_group(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 3>{});
        int buf = 0;
        auto body = [&](auto curc, auto tailc, int kt) {
            constexpr int cur = decltype(curc)::value;
            constexpr bool tail = decltype(tailc)::value;   // main iterations: more tiles to stage and to read, no branches in the body
            typedef std::integral_constant<int, cur ^ 1> Nxt;
            const int nbuf = buf + 1 == NST ? 0 : buf + 1;
            // tile kt+1 landed (its reads start below): tiles kt+2 .. kt+NST-1 may still be in flight; everybody's reads of tile kt
            // (issued one iteration ago) have returned before the barrier, so its slot can be refilled after it
            if (!tail) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * TP) : "memory");
            else if (kt + 1 < nk) {
                if (kt + NST - 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * TP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool more = !tail || kt + NST < nk, nxt = !tail || kt + 1 < nk;
            T* sbase = lds + buf * STAGE;
#define M3R_G96_GROUP(G)                                                                                                   \
            mma_group(curc, std::integral_constant<int, G>{});                                                              \
            if (more) glds16(src[G] + (kt + NST) * BK, sbase + (G * NW + wave) * RPP * BK);                                 \
            if (nxt) read_group(Nxt{}, nbuf, std::integral_constant<int, G>{});                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            M3R_G96_GROUP(0)
            M3R_G96_GROUP(1)
            M3R_G96_GROUP(2)
            M3R_G96_GROUP(3)
#undef M3R_G96_GROUP
            buf = nbuf;
        };
        int kt = 0;
        for (; kt + 1 + NST < nk; kt += 2) {   // both iterations stage a tile and read the next one
            body(std::integral_constant<int, 0>{}, std::false_type{}, kt);
            body(std::integral_constant<int, 1>{}, std::false_type{}, kt + 1);
        }
        for (; kt + 1 < nk; kt += 2) {
            body(std::integral_constant<int, 0>{}, std::true_type{}, kt);
            body(std::integral_constant<int, 1>{}, std::true_type{}, kt + 1);
        }
        if (kt < nk) body(std::integral_constant<int, 0>{}, std::true_type{}, kt);
    } else {
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) stage(t, t);
    if constexpr (LNF) {
        if (p.ln_stats != nullptr) ln_fold_rows<BM, 64 * NW>(p, m0, sm_ln, tid, n0 == 0);
    }
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt landed: at most NST-2 younger tiles x TP loads may still be in flight (tail: drain)
        if (kt + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * TP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const T* base = lds + buf * STAGE;
        v8 af[2][MF], wf[2][WS][NF];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int part = 0; part < WS; ++part)
#pragma unroll
                for (int j = 0; j < NF; ++j) wf[ks][part][j] = *reinterpret_cast<const v8*>(base + w_off[ks][part][j]);
#pragma unroll
            for (int i = 0; i < MF; ++i) af[ks][i] = *reinterpret_cast<const v8*>(base + a_off[ks][i]);
        }
        const int nt = kt + NST - 1;
        if (nt < nk) {
            int nb = buf + NST - 1;
            nb = nb >= NST ? nb - NST : nb;
            stage(nt, nb);                              // the buffer read in iteration kt-1
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int part = 0; part < WS; ++part)
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[i][j] = mfma16(wf[ks][part][j], af[ks][i], acc[i][j]);
        buf = buf + 1 == NST ? 0 : buf + 1;
    }
    }
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = m0 + wm * 32 + i * 16 + fr;
        if (m >= p.M) continue;
        f32x4 v[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) v[j] = acc[i][j];
        float mu = 0.f, rstd = 0.f;
        if constexpr (LNF) {
            if (p.ln_stats != nullptr) {
                mu = sm_ln[BM * 6 * 2 + (wm * 32 + i * 16 + fr) * 2];
                rstd = sm_ln[BM * 6 * 2 + (wm * 32 + i * 16 + fr) * 2 + 1];
            }
        }
        epilogue_row<T, EPI, NF>(p, outp, bias, m, n0 + wn * 32, fg, v, &pre[i], mu, rstd);
    }
}

template <class T, int EPI, int PF>
static int launch_96pf(const GemmArgs& a, hipStream_t s) {
    constexpr int NST = 4;
    const int nbn = a.N / 96, nbm = (a.M + 95) / 96;
    const size_t lds = (size_t)NST * (96 + 2 * 96) * 64 * sizeof(T) + ln_fold_lds_bytes<96, 576>();
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm96_kernel<T, EPI, NST, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm96_kernel<T, EPI, NST, PF>), dim3(nbm * nbn, a.batch > 1 ? a.batch : 1), dim3(576), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
template <class T, int EPI>
static int launch_96(const GemmArgs& a, hipStream_t s) {
    return launch_96pf<T, EPI, 1>(a, s);   // the software-pipelined K loop (the lock-step form measured 3 % slower: profiles/r02_gemm96_ab.txt)
}

// which kernel the last launch_gemm of this thread picked: "<family>/e<EPI>/w<WS>/n<BN>" = one symbol of a rocprofv3 kernel trace
// (must3r_hip_get_profile reports per-symbol times under these names so that every row can be matched with the trace)
static thread_local char g_gemm_pick[32] = "";
static void pick_name(const char* fam, int epi, int ws, int bn) { snprintf(g_gemm_pick, sizeof(g_gemm_pick), "%s/e%d/w%d/n%d", fam, epi, ws, bn); }
const char* gemm_last_kernel() { return g_gemm_pick; }

// Tile selection (measured on MI355X, scripts/bench_gemm.py): two resident blocks per CU beat every larger tile that
// leaves one (128x128 3-stage, 256x128 with 4 or 8 waves: 465-613 TF/s vs 644 TF/s on the scene's big-batch shapes).
//   plain weights : 128x128x64, 2 stages (64 KB)  for chip-filling grids, 64x64x64 4-stage ring (64 KB) otherwise
//   split weights : 128x64 (+64 lo) 2 stages (64 KB) / 64x64 (+64 lo) 3-stage ring (72 KB)
// K-tile depth 32 (the BK template parameter; 3-5 resident blocks per CU) was also measured: 605-626 TF/s plain,
// 397-419 vs 409 TF/s split -- no gain, so only BK = 64 is instantiated.
// minimum number of big tiles for the big-tile kernel (tunable for experiments: M3R_GEMM_MIN_BIG / _MIN_BIG_SPLIT)
static long min_big(bool split) { return split ? 384 : 192; }

// Launches of at most one 64 x 64 tile per CU (M = 768: proj, fc2, projq) run it with 8 waves (4 x 2, wave tile 16 x 32), a
// 6-8 slot LDS ring (one block per CU: the whole LDS can be prefetch depth) and the fragment reads of tile kt+1 issued
// before the MFMAs of tile kt (PIPE).  Measured at M = 768 with split weights: proj 8.7 -> 8.1 us, fc2 23.4 -> 20.7 us.
// What the experiments say about these launches (scripts/bench_gemm_small.py; debug builds that drop one component):
// per K-tile DMA, LDS reads and MFMAs each cost ~0.1 us ALONE (fc2: 16.3 / 16.8 / 16.6 us with one of them removed, 5.4 us
// with all three removed, 21 us with all) -- they add up instead of overlapping, whatever the ring depth (3 vs 6 slots:
// same), the bytes per tile (half the lo tile: -2 %) or the wave count; with two blocks per CU (qkv, fc1) the 4-wave form
// is as fast or faster; 64 x 32 tiles (twice the blocks, half the work each): same time per K-tile; the staggered
// two-group structure of gemm256_kernel on the 64 x 64 tile: same (fc2 20.4-21.0 us).  The slope is 0.35 us per
// 64-deep K-tile whatever the structure (proj 12 tiles 8.0 us, fc2 48 tiles 20.5 us); K off the power-of-two strides
// (3008 / 3136 instead of 3072) changes nothing, so it is not L2-channel camping either.  Open.
static bool small8(long tiles64) { return tiles64 <= 256; }
#ifndef SMALL_WGM
#define SMALL_WGM 4
#endif
#ifndef SMALL8_NST_SPLIT
#define SMALL8_NST_SPLIT 6   // 6 x 24 KB = 144 KB: one block per CU anyway, so the whole LDS can be prefetch depth
#endif
#ifndef SMALL8_NST_PLAIN
#define SMALL8_NST_PLAIN 8   // 8 x 16 KB = 128 KB
#endif

static bool use_48(const GemmArgs& a, long nb) {
    if (a.N % 48 || a.K % 64 || a.rope_tab != nullptr) return false;
    const long tiles = (long)((a.M + 47) / 48) * (a.N / 48) * nb;
    return tiles <= 256 && tiles >= 192;
}

// 96 x 96 tiles: split weights, one round of 224..256 tiles (M = 768: fc1 / feedback fc1 = 256 tiles; measured 17.8 -> 13.3 us);
static bool use_96(const GemmArgs& a, long nb) {
    if (a.N % 96 || a.K % 64) return false;
    const long tiles = (long)((a.M + 95) / 96) * (a.N / 96) * nb;
    return tiles <= 256 && tiles >= 224;   // (qkv at M = 768 is 192 tiles = 75 % of the CUs: measured 12.3 us vs 11.2 us on 64 x 64 tiles)
}

// M3R_GEMM256: 0 = never use the 8-wave kernel, 1 = by the fill rule below (default), 2 = whenever the shape allows it
static int gemm256_mode() { return opt(OPT_GEMM256); }
// the 8-wave kernel holds one block per CU: use it when its rounds over the 256 CUs are reasonably full
static int fill256(long tiles) {   // percentage of the CU slots of its rounds that do work
    const long rounds = (tiles + 255) / 256;
    return (int)(tiles * 100 / (rounds * 256));
}
static int g256k_mode() { return opt(OPT_G256K); }
// M3R_G256P (r05): the phase-staggered 64-deep kernel (gemm256p_kernel, two phases per K-tile, two barriers per phase) for the chip-filling launches.
//   plain weights  (M3R_G256P, default 1): 0 never (gemm256k_kernel), 1 every epilogue but the RoPE one (its instantiation spills), measured on the
//                  nine shapes of scripts/exp_gemm256.py: 2-6 % faster on each (profiles/r05_g256p_plain_ab.txt), 987 -> 1113 TF/s at K = 16384;
//   split weights  (M3R_G256P_SPLIT, default 1): 0 never, 1 where its 256 x 128 tiles fill their rounds at least as well as the widest tile gemm256_kernel
//                  would pick (decoder qkv N = 2304: 109 -> 103 us, K|V N = 1536: 72.6 -> 65.2 us; NOT the 256-wide encoder qkv 160 -> 162 us or the
//                  192-wide decoder proj 44.7 -> 51.8 us, profiles/r05_g256p_split_ab.txt), 2 whenever the shape allows.
// The other forms measured in r05: four phases per K-tile and one barrier per phase with group 1 lagging are template arguments of gemm256p_kernel the library does
// not instantiate; gemm256n (one phase per K-tile), gemm256pp (persistent tiles), gemm256q (one barrier per K-tile) and gemm256w (four waves, 128 x 128 wave tiles) live
// in scripts/probes/lab/*.inc and are compiled only with -DM3R_GEMM_LAB (scripts/probes/kloop_lab.hip).
static int g256p_mode(bool split) { return opt(split ? OPT_G256P_SPLIT : OPT_G256P); }
// (r04, measured and removed: two 256 x 128 blocks per CU -- the GELU launches' OCC = 2 form -- for the other split-weight epilogues, so that one
// block's fp32 read-modify-write epilogue runs under the other's K loop: RESID proj 44.7 -> 52.1 us (dec) / 67.2 -> 68.2 us (enc), fc2 122 -> 152 us,
// STORE16 K|V 71.9 -> 77.8 us, RoPE qkv +-1 %; nine split shapes 1246 -> 1308 us, step 497.6 -> 487.9 views/s.  profiles/r04_occ2_ab.txt.)
// (r04, measured and removed: the same two-blocks-per-CU form for the PLAIN-weight GELU launches instead of gemm256k -- enc fc1 146.8 -> 172.8 us, dec fc1 92.6 ->
// 105.4 us, step 495.9 -> 486.3 views/s: the 64-byte DMA rows and 256 x 128 tiles cost more than the hidden erf epilogue (~8 us of a 39 us tile) gives back.
// profiles/r04_plain_gelu_occ2.txt.)
// r06: does the default dispatch below send a chip-filling launch of this shape to gemm256s_kernel (split, sparse low part) / gemm256p_kernel<.., 1, 256> (plain)?
// Only then is the LN fold offered (kernels.hpp): forcing 256 x 256 tiles on a launch the fill rule gives to another tile would cost more than the fold returns.
bool gemm_fold256_shape_ok(int M, int N, int K, bool split) {
    if (M <= 0 || M % 256 != 0 || N % 256 != 0 || K % 64 != 0 || gemm256_mode() != 1) return false;
    const long rb256 = M / 256, t256 = rb256 * (N / 256), t128 = rb256 * (N / 128);
    if (t256 < 200 || fill256(t256) < 80) return false;
    if (!split) return g256p_mode(false) != 0;
    return g256p_mode(true) != 0 && opt(OPT_SPARSE_256) != 0 && opt(OPT_SPARSE_LO) != 0 && fill256(t256) + 12 >= fill256(t128);
}

template <class T, int EPI>
static int launch_epi(const GemmArgs& a, hipStream_t s, const char** err) {
    const long nb = a.batch > 1 ? a.batch : 1;
    if (a.fold256) {   // r06: the LN fold on the chip-filling tiles -- the caller asked gemm_fold256_shape_ok first; launch_gemm validated the operands
        int rc = 1;
        if constexpr (sizeof(T) == 2 && std::is_same<T, f16_t>::value) {
            if (a.wsplit == 2) {
                if constexpr (EPI == EPI_STORE16 || EPI == EPI_QKV_ROPE || EPI == EPI_RESID_F32) { pick_name("g256sf", EPI, 3, 256); rc = launch_256s<T, EPI>(a, s); }
            } else {
                if constexpr (EPI == EPI_STORE16_GELU || EPI == EPI_RESID_F32) { pick_name("g256pf", EPI, 1, 256); rc = launch_256p<T, EPI, 1, 256, 2>(a, s); }
            }
        }
        if (rc) *err = "gemm: fold256 is built for fp16: split STORE16 / QKV_ROPE / RESID_F32 and plain STORE16_GELU / RESID_F32";
        return rc;
    }
    // LN-fold producers (x16_out / copy32_out / stats_out) need a kernel whose epilogue carries the LN paths: the 64 x 64 / 96 / 48 tiles
    const bool lnp = a.x16_out != nullptr || a.copy32_out != nullptr || a.stats_out != nullptr;
    const int mode = lnp ? 0 : gemm256_mode();
    const long rb256 = (long)((a.M + 255) / 256) * nb;
    int rc;
    if (a.wsplit == 2) {
        if constexpr (sizeof(T) == 2 && std::is_same<T, f16_t>::value) {
            const long tiles = (long)((a.M + 127) / 128) * (a.N / 64) * nb;
            const long t256 = rb256 * (a.N / 256), t192 = rb256 * (a.N / 192), t128 = rb256 * (a.N / 128);
            const bool ok256 = a.N % 256 == 0 && a.K % 32 == 0, ok128 = a.N % 128 == 0 && a.K % 32 == 0;
            // 192-column tiles (wave tile 128 x 48): N = 768 at M = 15360 is 240 tiles = ONE round at 94 % fill, where 256 columns
            // give 180 tiles (70 %) and 128 columns two rounds.  Not for the RoPE epilogue (its rotate-half pairs need 32-column
            // aligned wave tiles).
            const bool ok192 = a.N % 192 == 0 && a.K % 32 == 0 && EPI != EPI_QKV_ROPE;
            int pick = 0;
            if (mode == 2) pick = !ok256 ? (ok128 ? 128 : 0) : 256;
            else if (mode == 1) {
                // cost ~ rounds over the 256 CUs x tile width; eligible when its rounds are reasonably full; ties -> wider tile
                long best = -1;
                const int bns[3] = {256, 192, 128};
                const long ts[3] = {t256, t192, t128};
                const bool oks[3] = {ok256, ok192, ok128};
                for (int i = 0; i < 3; ++i) {
                    if (!oks[i] || ts[i] < 200 || fill256(ts[i]) < 80) continue;
                    const long cost = ((ts[i] + 255) / 256) * bns[i];
                    if (best < 0 || cost < best) { best = cost; pick = bns[i]; }
                }
            }
            // fc1 (exact-erf GELU epilogue, ~17 VALU per output) of the chip-filling batches: two 256 x 128 blocks per CU, so that one block's
            // epilogue runs under the other's K loop.  Measured (r02, M = 15360): N = 3072, K = 768: 157 -> 143 us; N = 4096, K = 1024: 270 -> 265 us;
            // every other epilogue is faster with one block per CU and the deeper ring (qkv 186 vs 199-209 us, fc2 210 vs 228 us).
            constexpr bool gelu_occ2 = true;
            if (a.ln_stats != nullptr) {
                // LN-fold consumers: the kernels that carry the row-statistics prologue, whatever the tile count
                if constexpr (EPI == EPI_STORE16_GELU) { pick_name("g96", EPI, 2, 96); rc = a.N % 96 == 0 ? launch_96<T, EPI>(a, s) : 1; }
                else if constexpr (EPI == EPI_STORE16) { rc = a.N % 48 == 0 ? launch_48<T, EPI, 2>(a, s) : 1; pick_name(g_gemm48_bk == 128 ? "g48k128" : "g48", EPI, 2, 48); }
                else if constexpr (EPI == EPI_QKV_ROPE) { pick_name("g64", EPI, 2, 64); rc = launch_cfg<T, 64, 64, 2, 2, EPI, 3, 2>(a, s); }
                else rc = 1;
            } else if (EPI == EPI_STORE16_GELU && gelu_occ2 && mode != 0 && ok128 && t128 >= 1024) { pick_name("g256o2", EPI, 2, 128); rc = launch_256<T, EPI, 2, 128, 2>(a, s); }
            else if (EPI != EPI_HEAD && use_96(a, nb)) { pick_name("g96", EPI, 2, 96); rc = launch_96<T, EPI == EPI_HEAD ? EPI_STORE16 : EPI>(a, s); }
            else if (EPI != EPI_QKV_ROPE && EPI != EPI_HEAD && use_48(a, nb)) { rc = launch_48<T, EPI == EPI_QKV_ROPE || EPI == EPI_HEAD ? EPI_STORE16 : EPI, 2>(a, s); pick_name(g_gemm48_bk == 128 ? "g48k128" : "g48", EPI, 2, 48); }
            // r05: the 2:4-sparse low part (48 matrix instructions per K-tile instead of 64) wherever the launch fills the chip and the caller has the packed copy
            else if (a.Wlo_sp != nullptr && a.Widx_sp != nullptr && g256p_mode(true) != 0 && ok128 && a.K % 64 == 0 && t128 >= 200 && a.ln_stats == nullptr && !lnp &&
                     (fill256(t128) >= 80 || pick != 0)) {
                // 256 x 256 sparse tiles (48 matrix instructions per phase) where they fill their rounds about as well as the 256 x 128 ones (24 per phase, load-bound)
                const int s256 = opt(OPT_SPARSE_256);
                if (s256 && ok256 && a.K % 64 == 0 && t256 >= 200 && fill256(t256) + 12 >= fill256(t128)) { pick_name("g256s", EPI, 3, 256); rc = launch_256s<T, EPI>(a, s); }
                else { pick_name("g256ps", EPI, 3, 128); rc = launch_256p<T, EPI, 3, 128, 2>(a, s); }
            }
            else if (pick != 0 && ok128 && a.K % 64 == 0 && (g256p_mode(true) == 2 || (g256p_mode(true) == 1 && (pick == 128 || (pick == 192 && t128 >= 200 && fill256(t128) >= fill256(t192) && fill256(t128) >= 90))))) {
                pick_name("g256p", EPI, 2, 128); rc = launch_256p<T, EPI, 2, 128, 2>(a, s);
            }
            else if (pick != 0 && g256k_mode() >= 2 && ok128) { pick_name("g256k", EPI, 2, 128); rc = launch_256k<T, EPI, 2, 128>(a, s); }
            else if (pick == 256) { pick_name("g256", EPI, 2, 256); rc = launch_256<T, EPI, 2, 256>(a, s); }
            else if (pick == 192) { pick_name("g256", EPI, 2, 192); rc = launch_256<T, EPI == EPI_QKV_ROPE ? EPI_STORE16 : EPI, 2, 192>(a, s); }
            else if (pick == 128) { pick_name("g256", EPI, 2, 128); rc = launch_256<T, EPI, 2, 128>(a, s); }   // (two blocks per CU measured slower for every epilogue but the GELU one)
            else if (tiles >= min_big(true) && !lnp) { pick_name("g128", EPI, 2, 64); rc = launch_cfg<T, 128, 64, 2, 2, EPI, 2, 2>(a, s); }
            else if (small8((long)((a.M + 63) / 64) * (a.N / 64) * nb)) { pick_name("g64p", EPI, 2, 64); rc = launch_cfg<T, 64, 64, SMALL_WGM, 2, EPI, SMALL8_NST_SPLIT, 2, 64, 1>(a, s); }
            else { pick_name("g64", EPI, 2, 64); rc = launch_cfg<T, 64, 64, 2, 2, EPI, 3, 2>(a, s); }
        } else {
            *err = "gemm: split weights are only built for fp16";
            return 1;
        }
    } else {
        const bool n128 = (a.N % 128) == 0;
        const long tiles128 = (long)((a.M + 127) / 128) * (a.N / 128) * nb;
        const long t256 = rb256 * (a.N / 256);
        const bool ok256 = a.N % 256 == 0 && a.K % 32 == 0;
        if (a.ln_stats != nullptr) {
            // LN-fold consumers on plain weights (the Mlp fc1 of a one-view update in MUST3R_F16_WA mode): the 64 x 64 ring kernel
            if constexpr (EPI == EPI_STORE16_GELU || EPI == EPI_STORE16 || EPI == EPI_QKV_ROPE) { pick_name("g64", EPI, 1, 64); rc = launch_cfg<T, 64, 64, 2, 2, EPI, 4, 1>(a, s); }
            else rc = 1;
        } else if (EPI != EPI_QKV_ROPE && EPI != EPI_HEAD && sizeof(T) == 2 && use_48(a, nb)) {
            rc = launch_48<T, EPI == EPI_QKV_ROPE || EPI == EPI_HEAD ? EPI_STORE16 : EPI, 1>(a, s);
            pick_name(g_gemm48_bk == 128 ? "g48k128" : "g48", EPI, 1, 48);   // N = 768 one-view launches: 256 tiles of 48 x 48
        } else if (ok256 && a.K % 64 == 0 && g256p_mode(false) != 0 && EPI != EPI_QKV_ROPE && (mode == 2 || (mode == 1 && t256 >= 200 && fill256(t256) >= 80))) {
            pick_name("g256p", EPI, 1, 256); rc = launch_256p<T, EPI == EPI_QKV_ROPE ? EPI_STORE16 : EPI, 1, 256, 2>(a, s);
        } else if (ok256 && (mode == 2 || (mode == 1 && t256 >= 200 && fill256(t256) >= 80))) { pick_name(g256k_mode() >= 1 ? "g256k" : "g256", EPI, 1, 256); rc = g256k_mode() >= 1 ? launch_256k<T, EPI, 1, 256>(a, s) : launch_256<T, EPI, 1, 256>(a, s); }
        else if (n128 && tiles128 >= min_big(false) && !lnp) { pick_name("g128", EPI, 1, 128); rc = launch_cfg<T, 128, 128, 2, 2, EPI, 2, 1>(a, s); }
        else if (small8((long)((a.M + 63) / 64) * (a.N / 64) * nb)) { pick_name("g64p", EPI, 1, 64); rc = launch_cfg<T, 64, 64, SMALL_WGM, 2, EPI, SMALL8_NST_PLAIN, 1, 64, 1>(a, s); }
        else { pick_name("g64", EPI, 1, 64); rc = launch_cfg<T, 64, 64, 2, 2, EPI, 4, 1>(a, s); }
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
    if ((epi == EPI_STORE16 || epi == EPI_STORE16_GELU || epi == EPI_QKV_ROPE) && a.ldc % 8 != 0) {   // 16-byte stores of 8 outputs
        *err = "gemm: 16-bit outputs need ldc % 8 == 0"; return 1;
    }
    if (epi == EPI_QKV_ROPE && (a.pos == nullptr || a.rope_tab == nullptr || a.rope_cols % 64 != 0)) {
        *err = "gemm: rope epilogue needs pos, table and 64-aligned rope_cols"; return 1;
    }
    if (epi == EPI_HEAD && (a.N % 112 != 0 || a.ntok <= 0 || a.gw <= 0)) { *err = "gemm: head epilogue geometry"; return 1; }
    if (a.fold256) {
        const bool cons = epi == EPI_STORE16 || epi == EPI_STORE16_GELU || epi == EPI_QKV_ROPE;
        if (dt != DT_F16 || a.batch > 1 || a.N % 256 != 0 || a.K % 64 != 0 || (!cons && epi != EPI_RESID_F32)) { *err = "gemm: fold256 needs fp16, one problem, N % 256 == 0"; return 1; }
        if (cons && (a.ln_stats == nullptr || a.ln_s == nullptr || a.bias == nullptr || a.ln_shift == nullptr || (a.K != 768 && a.K != 1024) || a.x16_out || a.stats_out || a.copy32_out)) {
            *err = "gemm: fold256 consumer needs ln_stats, ln_s, bias, ln_shift and K = 768 or 1024"; return 1;
        }
        if (!cons && (a.x16_out == nullptr || a.stats_out == nullptr || a.M % 256 != 0 || a.ln_stats != nullptr)) { *err = "gemm: fold256 producer needs x16_out, stats_out, M % 256 == 0"; return 1; }
        if (a.wsplit == 2 && (a.Wlo_sp == nullptr || a.Widx_sp == nullptr)) { *err = "gemm: fold256 on split weights needs the packed sparse low part"; return 1; }
    } else
    if (a.ln_stats != nullptr) {
        const bool epi_ok = epi == EPI_STORE16 || epi == EPI_STORE16_GELU || epi == EPI_QKV_ROPE;
        if (!epi_ok || a.ln_s == nullptr || (a.wsplit != 2 && a.wsplit != 0) || dt != DT_F16 || a.K != 16 * LNF_SLOTS || (a.batch > 1)) {
            *err = "gemm: LN fold needs a 16-bit-store epilogue, ln_s, fp16 weights, K = 768, no batch";
            return 1;
        }
    }
    if ((a.x16_out || a.copy32_out || a.stats_out) && !(epi == EPI_RESID_F32 || epi == EPI_F32)) {
        *err = "gemm: LN-fold outputs need the RESID_F32 / F32 epilogue";
        return 1;
    }
    // the consumer of the statistics reads LNF_SLOTS fragments per row (its K = 768): a producer of another width would pair with misaligned rows
    if (!a.fold256 && a.stats_out && a.N != 16 * LNF_SLOTS) { *err = "gemm: stats_out (LN-fold producer) needs N = 768"; return 1; }
    return dt == DT_BF16 ? launch_t<bf16_t>(epi, a, s, err) : launch_t<f16_t>(epi, a, s, err);
}

}  // namespace m3r
