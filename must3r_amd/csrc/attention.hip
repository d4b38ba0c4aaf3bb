// Flash-style softmax attention for gfx950, head dim 64, 16-bit operands / fp32 softmax + accumulation.
//
// Transposed formulation so that nothing ever moves across lanes except the row max / row sum:
//     S^T[key][q] = K . Q^T        (MFMA A = K tile rows from LDS,       B = Q fragments held in VGPRs)
//     O^T[d][q]   = V^T . P^T      (MFMA A = V^T via ds_read_b64_tr_b16, B = P^T = exp2(S^T - m) in place)
// With v_mfma_f32_16x16x32 the C layout of S^T (lane: q = l&15, keys 4*(l>>4)+r) is exactly the B-operand
// layout the second product needs (k-slot (l>>4)*8+e <-> keys {4g+e, 16+4g+e}), and V^T is read from the
// row-major V tile with the hardware transposing LDS read, keyed the same way.  q is per-lane in both
// accumulators, so the online-softmax rescale is a per-lane multiply.
//
// Block = 4 waves x 32 query rows; K/V tiles of 64 keys DMA'd HBM->LDS (global_load_lds), double
// buffered, one barrier per tile.  Keys may exclude one contiguous range per view (MUSt3R own-token rule,
// decoder.py:119-139): fully excluded tiles are never loaded, partially excluded ones are masked.
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"
#include "options.hpp"

namespace m3r {

constexpr int ATT_QW = 32;               // query rows per wave (template default); 16 for launches too small to fill the chip
constexpr int ATT_QB = 4 * ATT_QW;       // per block (default geometry, used by the split heuristic)
constexpr int ATT_KT = 64;               // keys per tile
constexpr float ATT_THR = 6.0f;          // lazy-rescale threshold in log2 units (P <= 64)
constexpr float ATT_LIM = 4096.0f;       // LZ = 1: largest row sum of ONE 64-key tile that does not move the references (64 keys x 2^ATT_THR)

// blockIdx -> (group = view x head, key split, query block).  The blocks that share K/V tiles are the query blocks of one
// (group, split) PAIR; a pair stays on one XCD (blockIdx % 8: observed placement, speed only) so its tiles are L2 hits, and the
// pairs are dealt round-robin over the XCDs.  (Round 1 dealt whole groups: the 12 heads of a one-view launch -- every cross /
// self attention of the memory update -- gave XCDs 0-3 two heads and XCDs 4-7 one, i.e. a third of the chip idle for half the
// kernel.)  When even the pairs do not divide evenly (12 heads, no split: the one-view self attention, whose K/V is 196 KB
// per head anyway) the blocks themselves are dealt round-robin: nqb < 0 on entry selects that map.
__device__ __forceinline__ bool attn_block_coords(int nqb_signed, int ngrp, int nsplit, int& grp, int& split, int& qb) {
    const int npairs = ngrp * nsplit;
    int pair;
    if (nqb_signed < 0) {
        const int nqb = -nqb_signed;
        qb = blockIdx.x % nqb;
        pair = blockIdx.x / nqb;
    } else {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        qb = slot % nqb_signed;
        pair = (slot / nqb_signed) * 8 + xcd;
    }
    if (pair >= npairs) return false;
    grp = pair / nsplit;
    split = pair - grp * nsplit;
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// attn3_kernel: the round-1 kernel (16 x 16 tiles, removed in r03) with the per-tile instruction count cut down.  The r02 cycle trace (scripts/probes/attn_trace.hip)
// and the QW / pipelining experiments say that a SIMD spends ~1500 cycles per (wave, key tile) however many waves share it and
// however the work is arranged: 576 of them are the 36 MFMAs, the rest is the ~265 other instructions the loop body issues
// (177 VALU, 28 LDS / DMA, ~60 scalar) -- the kernel is bound by what it ISSUES, and most of that was bookkeeping:
//   * K/V staging: `buffer_load_dwordx4 ... lds` with a per-lane voffset computed once and the tile position in the SCALAR
//     offset: no per-tile 64-bit address arithmetic, no clamping (rows past nk are out of the descriptor's range and read as
//     zero; they are masked like before).
//   * LDS addresses: the swizzles are XORs with per-lane constants, so every fragment address is one of a few per-lane VGPRs
//     plus an immediate (buffer, k-slot, row block): 2 VGPRs for the K fragments, 4 for the transposing V reads; the loop is
//     unrolled over the two buffers so that the buffer is part of the immediate.
//   * (r03, measured and dropped: the wave's four DMA pieces of the next tile issued between the MFMA groups of the tile instead of as a burst
//     behind the barrier: render CA 905 vs 919 TF/s, one-view split launches -4 %.)
//   * softmax fast path: the cross-lane row maximum is only needed when the reference moves.  Whether it has to move is decided
//     from the PER-LANE maxima (`any lane above the threshold`, one compare per 16 queries), the permlane exchanges, the
//     first-tile selects and the rescale live in one wave-uniform slow block that updates S, O, the row sums and m IN PLACE
//     (no second register version of the accumulators, no copies at a join).
// Arithmetic: S - m from the accumulator init, exp2, P rounded to T, row sums from the ones MFMA; the slow path subtracts the
// shift before the exp2.
//   * r05 (LZ = 1, the default): the fast path computes NO maxima.  The `ones` MFMAs run on a fresh accumulator BEFORE the P.V products, so the tile's
//     row sums are in hand when the products start: a sum <= ATT_LIM = 4096 proves that no score of the tile is more than 12 log2 units above its row's
//     reference (P <= 4096 is exact business for a 16-bit P and an fp32 accumulator); a larger sum, an infinity (a score 16+ above the reference overflows
//     fp16, 128+ the exp2 itself) or a NaN sends the wave through the slow block, which reads the K tile again (it is still in LDS) and moves the
//     references as before.  The 19 max3 / max instructions per tile become 2 compares + 2 adds; 75 -> 52 VALU instructions per 36 MFMAs.  Measured
//     (profiles/r05_attn_*): render cross attention +1-3 %, the 28-scene step +1.1 %.  (The per-pair exp2 on packed fp16 that VERDICT r04 proposed was
//     costed first: there is no packed transcendental, and a 2^x on v_pk_fma_f16 -- split, degree-3 polynomial, exponent insertion, clamp -- is >= 9 packed
//     instructions per PAIR against 2 v_exp_f32 + 1 v_cvt_pk: more issue slots, not fewer.  DESIGN.md section 3.2.)
template <int O0, int O1>
__device__ __forceinline__ void lds_tr_x8_imm(unsigned a0, unsigned a1, unsigned a2, unsigned a3, u32x2 (&r)[8]) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
        "ds_read_b64_tr_b16 %2, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
        "ds_read_b64_tr_b16 %4, %10 offset:%12\n\t"
        "ds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
        "ds_read_b64_tr_b16 %6, %11 offset:%12\n\t"
        "ds_read_b64_tr_b16 %7, %11 offset:%13\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(O0), "n"(O1)
        : "memory");
}

// maximum of 16 floats: four independent v_max3 chains merged at the end (depth 4, 8 instructions, ONE asm statement: hipcc pads
// separate asm statements with s_nop and fmaxf chains with NaN-quieting v_max x, x, x -- both are issue slots the loop cannot afford)
__device__ __forceinline__ float max16f(const f32x4& a, const f32x4& b, const f32x4& c, const f32x4& d) {
    float r, t1, t2, t3;
    asm("v_max3_f32 %0, %4, %5, %6\n\t"
        "v_max3_f32 %1, %7, %8, %9\n\t"
        "v_max3_f32 %2, %10, %11, %12\n\t"
        "v_max3_f32 %3, %13, %14, %15\n\t"
        "v_max3_f32 %0, %0, %16, %17\n\t"
        "v_max3_f32 %1, %1, %18, %19\n\t"
        "v_max3_f32 %0, %0, %1, %2\n\t"
        "v_max_f32 %0, %0, %3"
        : "=&v"(r), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]),
          "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]));
    return r;
}

template <int V> __device__ __forceinline__ std::integral_constant<int, V> to_constant(std::integral_constant<int, V>) { return {}; }
__device__ __forceinline__ std::integral_constant<int, 0> to_constant(int) { return {}; }
template <class T, int QW, int ABL = 0, int LZ = 0, int NB = 2, int PR = 0>   // PR: s_setprio experiment (1: MFMA clusters raised, 2: the exp2 cluster raised); ABL: timing ablations for experiments (1 no exp2, 2 no max / slow path, 3 no P.V MFMAs, 4 no cvt, 5 no DMA in the loop,
                                         // 6 no barrier / vmcnt wait, 7 = 5 + 6, 8 = 7 + no LDS reads, 9 = 8 + no softmax VALU: MFMAs only)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QW == 32 ? 3 : 2))) attn3_kernel(const AttnArgs p, const int nqb, const int ngrp, const int nsplit) {
    typedef typename Vec<T>::v8 v8;
    typedef typename Vec<T>::v4 v4;
    constexpr int QF = QW / 16;
    constexpr int QB = 4 * QW;
    constexpr int TILE = ATT_KT * 64;                                    // elements of one K (or V) tile
    __shared__ __attribute__((aligned(16))) T smem_kv[NB][2][TILE];      // [buffer][K|V][64 keys x 64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;

    int grp, split, qb;
    if (!attn_block_coords(nqb, ngrp, nsplit, grp, split, qb)) return;
    const int view = grp / p.heads, head = grp - view * p.heads;
    const AttnView vw = p.view0_inline ? p.view0 : p.views[view];
    if (qb * QB >= vw.nq) return;

    const T* __restrict__ Q = reinterpret_cast<const T*>(p.Q);
    const T* __restrict__ K = reinterpret_cast<const T*>(p.K) + (size_t)vw.kv_row0 * p.ldk + head * 64;
    const T* __restrict__ V = reinterpret_cast<const T*>(p.V) + (size_t)vw.kv_row0 * p.ldv + head * 64;

    const int qr0 = qb * QB + wave * QW;
    v8 qf_[QF][2];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        int r = qr0 + f * 16 + fr;
        r = r < vw.nq ? r : vw.nq - 1;
        const T* src = Q + (size_t)(vw.q_row0 + r) * p.ldq + head * 64 + fg * 8;
        qf_[f][0] = *reinterpret_cast<const v8*>(src);
        qf_[f][1] = *reinterpret_cast<const v8*>(src + 32);
    }

    const int nk = vw.nk, slo = vw.skip_lo, shi = vw.skip_hi;
    const int ntiles = (nk + ATT_KT - 1) / ATT_KT;
    auto fully_skipped = [&](int t) {
        const int k0 = t * ATT_KT;
        const int k1 = (k0 + ATT_KT < nk) ? k0 + ATT_KT : nk;
        return k0 >= slo && k1 <= shi;
    };
    const int tps = (ntiles + nsplit - 1) / nsplit;
    const int t_begin = split * tps;
    const int t_end = (t_begin + tps < ntiles) ? t_begin + tps : ntiles;
    auto advance = [&](int t) {
        ++t;
        if (shi > slo)
            while (t < t_end && fully_skipped(t)) ++t;
        return t;
    };

    // ---- staging through buffer descriptors: rows >= nk lie outside the descriptor and read as zero
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(K), 0, ((nk - 1) * p.ldk + 64) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(V), 0, ((nk - 1) * p.ldv + 64) * 2, 0x00020000);
    int vok[2], vov[2];
    {
        const int srow = lane >> 3, pch = lane & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (wave * 2 + i) * 8 + srow;
            vok[i] = (r * p.ldk + swz(r, pch) * 8) * 2;
            vov[i] = (r * p.ldv + swz_v(r, pch) * 8) * 2;
        }
    }
    const int tstride_k = ATT_KT * p.ldk * 2, tstride_v = ATT_KT * p.ldv * 2;
    // bufc: the buffer as a compile-time constant (NB = 2: the loop is unrolled over the two buffers and the buffer is part of every LDS immediate) or as a
    // wave-uniform run-time index (NB = 3: one loop body, six address additions per tile)
    auto stage = [&](int t, auto bufc) {
        const int buf = bufc;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)&smem_kv[buf][0][(wave * 2 + i) * 8 * 64], 16, vok[i],
                                                     t * tstride_k, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)&smem_kv[buf][1][(wave * 2 + i) * 8 * 64], 16, vov[i],
                                                     t * tstride_v, 0, 0);
        }
    };

    // ---- per-lane LDS offsets (elements for the K fragments, byte addresses for the transposing V reads)
    //   K fragment (ks, kf): row 16 kf + fr, chunk (4 ks + fg) ^ ((fr >> 1) & 7) = z ^ (4 ks)
    //   V^T read (ks, d, half h): row 32 ks + 16 h + 4 fg + (fr >> 2), chunk pair (d ^ x), x = (2 fg + (fr >> 3)) & 3
    int koff[2];
    {
        const int z = fg ^ ((fr >> 1) & 7);
        koff[0] = fr * 64 + z * 8;
        koff[1] = fr * 64 + (z ^ 4) * 8;
    }
    unsigned vaddr[4];
    {
        const int x = (2 * fg + (fr >> 3)) & 3;
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const T*)(&smem_kv[0][1][0]);
        const unsigned lane_base = (unsigned)((4 * fg + (fr >> 2)) * 128 + ((fr >> 1) & 1) * 16 + (fr & 1) * 8);
#pragma unroll
        for (int d = 0; d < 4; ++d) vaddr[d] = base + lane_base + (unsigned)((d ^ x) * 32);
    }

    // nm_[f] = -m/c splat: the C operand of the first S MFMA of every key fragment (S - m for free); it only changes in the slow
    // block.  any_first: some query of this wave has not seen a valid key yet (wave-uniform, re-evaluated in the slow block).
    f32x4 o_[4][QF], ol_[QF], nm_[QF];
    float m_[QF];
    bool any_first = true;
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        m_[f] = -INFINITY;
        ol_[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        nm_[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 4; ++d) o_[d][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool has_skip = shi > slo;
    const float c = p.q_prescaled ? 1.0f : p.scale * 1.44269504088896340736f;
    const float inv_c = 1.0f / c;
    v8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;
    if constexpr (LZ != 0) asm volatile("" : "+v"(ones));   // kept in four VGPRs (rebuilt from scalar registers on every tile otherwise: two v_mov_b64)

    // ---- S^T = K Q^T - m (C operand of the first MFMA), scaled, masked
    auto scores = [&](f32x4 (&s_)[4][QF], int t, auto bufc) {
        const int buf = bufc;
        const T* k_ = smem_kv[buf][0];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            const v8 kfrag = ABL >= 8 ? qf_[0][1] : *reinterpret_cast<const v8*>(k_ + koff[0] + kf * 16 * 64);
#pragma unroll
            for (int f = 0; f < QF; ++f) s_[kf][f] = mfma16(kfrag, qf_[f][0], nm_[f]);
        }
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            const v8 kfrag = ABL >= 8 ? qf_[0][0] : *reinterpret_cast<const v8*>(k_ + koff[1] + kf * 16 * 64);
#pragma unroll
            for (int f = 0; f < QF; ++f) s_[kf][f] = mfma16(kfrag, qf_[f][1], s_[kf][f]);
        }
        if (!p.q_prescaled) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int f = 0; f < QF; ++f) s_[kf][f] *= c;
        }
        const int k0 = t * ATT_KT;
        const bool need_mask = (k0 + ATT_KT > nk) || (has_skip && k0 < shi && k0 + ATT_KT > slo);
        if (need_mask) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kf * 16 + fg * 4 + r;
                    const bool bad = (key >= nk) | ((key >= slo) & (key < shi));
                    const float pen = bad ? -INFINITY : 0.f;
#pragma unroll
                    for (int f = 0; f < QF; ++f) s_[kf][f][r] += pen;
                }
        }
    };
    // the wave-uniform slow block: every reference that has to move moves, S, O, the row sums and m updated IN PLACE
    auto move_references = [&](f32x4 (&s_)[4][QF]) {
        bool fst = false;
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float mx = fmaxf(s_[0][f][0], s_[0][f][1]);
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = (kf == 0 ? 2 : 0); r < 4; ++r) mx = fmaxf(mx, s_[kf][f][r]);
            mx = quad_row_max(mx);
            const bool first = (m_[f] == -INFINITY);
            float d = first ? mx : fmaxf(mx, 0.f);
            d = (d == -INFINITY) ? 0.f : d;                       // row still has no valid key
            const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) s_[kf][f] -= d;
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) o_[dd][f] *= alpha;
            if constexpr (LZ == 0) ol_[f] *= alpha; else ol_[f][0] *= alpha;
            m_[f] = first ? ((mx == -INFINITY) ? -INFINITY : d) : m_[f] + d;
            const float nm = (m_[f] == -INFINITY) ? 0.f : -m_[f] * inv_c;
            nm_[f] = f32x4{nm, nm, nm, nm};
            fst |= (m_[f] == -INFINITY);
        }
        any_first = __any(fst);
    };
    // P^T = exp2(S^T - m) rounded to T, as the B fragments of the second product: k-slot e of lane group g <-> keys {4g+e (e<4), 16+4g+e-4 (e>=4)}
    // of the 32-key slot ks
    auto probabilities = [&](f32x4 (&s_)[4][QF], v8 (&pb)[2][QF]) {
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int f = 0; f < QF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_[kf][f][r] = ABL == 9 ? s_[kf][f][r] : (ABL == 1 ? s_[kf][f][r] * 0.001f : __builtin_amdgcn_exp2f(s_[kf][f][r]));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                f32x8 pv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[r] = s_[2 * ks][f][r];
                    pv[4 + r] = s_[2 * ks + 1][f][r];
                }
                if (ABL == 4 || ABL == 9) __builtin_memcpy(&pb[ks][f], &pv, 16);   // first four fp32 words as the fragment: no conversion
                else pb[ks][f] = cvt8<T>(pv);
            }
    };
    auto value_product = [&](const v8 (&pb)[2][QF], auto bufc) {
        constexpr bool rt = std::is_same<decltype(bufc), int>::value;
        constexpr int buf = rt ? 0 : (int)decltype(to_constant(bufc))::value;
        unsigned va[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) va[d] = rt ? vaddr[d] + (unsigned)((int)bufc * 4 * TILE) : vaddr[d];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ABL == 3) { asm volatile("" ::"v"(pb[ks][0])); continue; }
            u32x2 tr[8];
            if (ABL >= 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) __builtin_memcpy(&tr[e], &qf_[0][e & 1], 8);
            } else if (ks == 0) lds_tr_x8_imm<buf * 4 * TILE, buf * 4 * TILE + 16 * 128>(va[0], va[1], va[2], va[3], tr);
            else lds_tr_x8_imm<buf * 4 * TILE + 32 * 128, buf * 4 * TILE + 48 * 128>(va[0], va[1], va[2], va[3], tr);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                v4 lo, hi;
                __builtin_memcpy(&lo, &tr[2 * d], 8);
                __builtin_memcpy(&hi, &tr[2 * d + 1], 8);
                const v8 vfrag = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int f = 0; f < QF; ++f) o_[d][f] = mfma16(vfrag, pb[ks][f], o_[d][f]);
            }
        }
    };

    // ---- LZ = 2 (r05, experiment builds only: measured SLOWER, profiles/r05_attn_halves_*.txt): the two 16-query halves of the wave one stage apart, so that every exp2 / cvt cluster has an INDEPENDENT MFMA cluster next to it in the
    // same basic block (the S product of the other half, or the P.V product of the other half) -- inside a wave, behind in-order issue, softmax VALU work only
    // overlaps matrix work that does not depend on it:
    //     [K fragments]  S(0)  |  S(1) || P(0), sums(0), check(0)  |  [V fragments]  O(0) += V P(0) || P(1), sums(1), check(1)  |  O(1) += V P(1)
    // K and V fragments are read once per tile and serve both halves from registers.  The row-sum rule of LZ = 1 applies per half.
    bool first_h[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) first_h[f] = true;
    auto scores_half = [&](f32x4 (&sh)[4], const v8 (&kfr)[2][4], int f) {
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) sh[kf] = mfma16(kfr[0][kf], qf_[f][0], nm_[f]);
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) sh[kf] = mfma16(kfr[1][kf], qf_[f][1], sh[kf]);
    };
    auto read_k = [&](v8 (&kfr)[2][4], auto bufc) {
        const int buf = bufc;
        const T* k_ = smem_kv[buf][0];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) kfr[ks][kf] = *reinterpret_cast<const v8*>(k_ + koff[ks] + kf * 16 * 64);
    };
    auto scale_mask_half = [&](f32x4 (&sh)[4], int t) {   // the rare part of a score tile: unscaled q (op entry), partial / excluded key tiles
        if (!p.q_prescaled) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) sh[kf] *= c;
        }
        const int k0 = t * ATT_KT;
        if ((k0 + ATT_KT > nk) || (has_skip && k0 < shi && k0 + ATT_KT > slo)) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kf * 16 + fg * 4 + r;
                    const bool bad = (key >= nk) | ((key >= slo) & (key < shi));
                    sh[kf][r] += bad ? -INFINITY : 0.f;
                }
        }
    };
    auto probs_half = [&](f32x4 (&sh)[4], v8 (&pbh)[2], f32x4& lth) {
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) sh[kf][r] = __builtin_amdgcn_exp2f(sh[kf][r]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f32x8 pv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[r] = sh[2 * ks][r];
                pv[4 + r] = sh[2 * ks + 1][r];
            }
            pbh[ks] = cvt8<T>(pv);
        }
        lth = mfma16(ones, pbh[0], f32x4{0.f, 0.f, 0.f, 0.f});
        lth = mfma16(ones, pbh[1], lth);
    };
    auto move_half = [&](f32x4 (&sh)[4], int f) {
        float mx = fmaxf(sh[0][0], sh[0][1]);
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = (kf == 0 ? 2 : 0); r < 4; ++r) mx = fmaxf(mx, sh[kf][r]);
        mx = quad_row_max(mx);
        const bool first = (m_[f] == -INFINITY);
        float d = first ? mx : fmaxf(mx, 0.f);
        d = (d == -INFINITY) ? 0.f : d;                       // row still has no valid key
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-d);
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) sh[kf] -= d;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) o_[dd][f] *= alpha;
        ol_[f][0] *= alpha;
        m_[f] = first ? ((mx == -INFINITY) ? -INFINITY : d) : m_[f] + d;
        const float nm = (m_[f] == -INFINITY) ? 0.f : -m_[f] * inv_c;
        nm_[f] = f32x4{nm, nm, nm, nm};
        first_h[f] = __any(m_[f] == -INFINITY);
    };
    // the half's P fragments stand, or the wave goes through the slow block: S of this half again from the K tile in LDS, references moved, P and sums again
    auto settle_half = [&](f32x4 (&sh)[4], v8 (&pbh)[2], f32x4& lth, int f, int t, auto bufc) {
        if (first_h[f] || __any(!(lth[0] <= ATT_LIM))) {
            asm volatile("" ::: "memory");
            v8 kfr[2][4];
            read_k(kfr, bufc);
            scores_half(sh, kfr, f);
            scale_mask_half(sh, t);
            move_half(sh, f);
            probs_half(sh, pbh, lth);
        }
        ol_[f][0] += lth[0];
    };

    auto compute = [&](int t, auto bufc) {
        f32x4 s_[4][QF];
        if constexpr (LZ == 2) {
            static_assert(LZ != 2 || QF == 2, "skewed halves: 32 query rows per wave");
            constexpr bool rt = std::is_same<decltype(bufc), int>::value;
            constexpr int buf = rt ? 0 : (int)decltype(to_constant(bufc))::value;
            const int k0 = t * ATT_KT;
            const bool rare = !p.q_prescaled || (k0 + ATT_KT > nk) || (has_skip && k0 < shi && k0 + ATT_KT > slo);
            f32x4 s0[4], s1[4], lt0, lt1;
            v8 pb0[2], pb1[2];
            {
                v8 kfr[2][4];
                read_k(kfr, bufc);
                scores_half(s0, kfr, 0);
                if (rare) scale_mask_half(s0, t);
                scores_half(s1, kfr, 1);            // || the exp2 / cvt of half 0
                probs_half(s0, pb0, lt0);
            }
            settle_half(s0, pb0, lt0, 0, t, bufc);
            if (rare) scale_mask_half(s1, t);
            u32x2 tr[2][8];
            lds_tr_x8_imm<buf * 4 * TILE, buf * 4 * TILE + 16 * 128>(vaddr[0], vaddr[1], vaddr[2], vaddr[3], tr[0]);
            lds_tr_x8_imm<buf * 4 * TILE + 32 * 128, buf * 4 * TILE + 48 * 128>(vaddr[0], vaddr[1], vaddr[2], vaddr[3], tr[1]);
            v8 vfr[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    v4 lo, hi;
                    __builtin_memcpy(&lo, &tr[ks][2 * d], 8);
                    __builtin_memcpy(&hi, &tr[ks][2 * d + 1], 8);
                    vfr[ks][d] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int d = 0; d < 4; ++d) o_[d][0] = mfma16(vfr[ks][d], pb0[ks], o_[d][0]);      // || the exp2 / cvt of half 1
            probs_half(s1, pb1, lt1);
            settle_half(s1, pb1, lt1, 1, t, bufc);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int d = 0; d < 4; ++d) o_[d][1] = mfma16(vfr[ks][d], pb1[ks], o_[d][1]);
        } else if constexpr (LZ == 0) {
            scores(s_, t, bufc);
            // ---- does any reference have to move?  decided from ONE per-lane maximum over all the lane's scores
            float mxa = -INFINITY;
            if (ABL != 2 && ABL != 9) {
#pragma unroll
                for (int f = 0; f < QF; ++f) {
                    const float mf = max16f(s_[0][f], s_[1][f], s_[2][f], s_[3][f]);
                    mxa = f == 0 ? mf : fmaxf(mxa, mf);
                }
            }
            if (ABL != 2 && ABL != 9 && (any_first || __any(mxa > ATT_THR))) move_references(s_);   // wave-uniform, rare after the first tile
            v8 pb[2][QF];
            probabilities(s_, pb);
            if (ABL != 3) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int f = 0; f < QF; ++f) ol_[f] = mfma16(ones, pb[ks][f], ol_[f]);
            }
            value_product(pb, bufc);
        } else {
            // ---- r05: no maxima in the fast path.  The tile's row sums (the `ones` MFMA on a fresh accumulator) bound every P of the tile: a sum
            // <= ATT_LIM means no score is more than log2(ATT_LIM) above its row's reference, P fits T and the products carry on; a larger one
            // (or an infinity / NaN out of the exp2 or the rounding) sends the wave through the slow block, which rebuilds S from the K tile that is still
            // in LDS.  19 VALU instructions per tile (the max3 trees) become 4.
            v8 pb[2][QF];
            f32x4 lt[QF];
            auto tile_sums = [&]() {
#pragma unroll
                for (int f = 0; f < QF; ++f) lt[f] = mfma16(ones, pb[0][f], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int f = 0; f < QF; ++f) lt[f] = mfma16(ones, pb[1][f], lt[f]);
            };
            if constexpr (PR == 1) __builtin_amdgcn_s_setprio(1);
            scores(s_, t, bufc);
            if constexpr (PR == 1) __builtin_amdgcn_s_setprio(0);
            bool slow = any_first;
            if (!slow) {
                if constexpr (PR == 2) __builtin_amdgcn_s_setprio(1);
                probabilities(s_, pb);
                if constexpr (PR == 2) __builtin_amdgcn_s_setprio(0);
                tile_sums();
                bool over = false;
#pragma unroll
                for (int f = 0; f < QF; ++f) over |= !(lt[f][0] <= ATT_LIM);
                slow = __any(over);
                if (slow) {
                    asm volatile("" ::: "memory");   // the K fragments are READ AGAIN (kept in registers across the fast path they would cost it 32 VGPRs)
                    scores(s_, t, bufc);
                }
            }
            if (slow) {
                move_references(s_);
                probabilities(s_, pb);
                tile_sums();
            }
#pragma unroll
            for (int f = 0; f < QF; ++f) ol_[f][0] += lt[f][0];
            if constexpr (PR == 1) __builtin_amdgcn_s_setprio(1);
            value_product(pb, bufc);
            if constexpr (PR == 1) __builtin_amdgcn_s_setprio(0);
        }
    };

    if constexpr (NB == 3) {
        // r05: three buffers, the DMA of a tile issued TWO tiles ahead of its use (one tile ahead, a block's ~1.5 us tile period does not cover the latency of a
        // tile fetched under load and every wave of the block waits it out at the barrier).  Loads return in order: four pieces per wave and tile, so
        // vmcnt(4) = "tile t has landed, tile t + 1 may still be in flight".  The buffer restaged behind the barrier of tile t is the one tile t - 1 was read from.
        int t = advance(t_begin - 1), tn = t_end;
        if (t < t_end) {
            stage(t, 0);
            tn = advance(t);
            if (tn < t_end) stage(tn, 1);
        }
        int b = 0;
#pragma clang loop unroll(disable)
        while (t < t_end) {
            if (tn < t_end) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            const int tnn = tn < t_end ? advance(tn) : t_end;
            const int b2 = b == 0 ? 2 : b - 1;
            if (tnn < t_end) stage(tnn, b2);
            compute(t, b);
            t = tn;
            tn = tnn;
            b = b == 2 ? 0 : b + 1;
        }
    } else {
        int t = advance(t_begin - 1);
        if (t < t_end) stage(t, std::integral_constant<int, 0>{});
        while (t < t_end) {
            constexpr bool no_dma = ABL == 5 || ABL >= 7, no_bar = ABL >= 6;
            {
                if (!no_bar) {
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                    __syncthreads();
                }
                const int tn = advance(t);
                if (tn < t_end && !no_dma) stage(tn, std::integral_constant<int, 1>{});
                compute(t, std::integral_constant<int, 0>{});
                t = tn;
            }
            if (t >= t_end) break;
            {
                if (!no_bar) {
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                    __syncthreads();
                }
                const int tn = advance(t);
                if (tn < t_end && !no_dma) stage(tn, std::integral_constant<int, 0>{});
                compute(t, std::integral_constant<int, 1>{});
                t = tn;
            }
        }

    }

    // ---- normalise and store: lane (q = fr, g) holds O[q][d = 16 dd + 4 g + r]
    T* __restrict__ O = reinterpret_cast<T*>(p.O);
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const float l = ol_[f][0];
        const int q = qr0 + f * 16 + fr;
        if (q >= vw.nq) continue;
        const size_t row = (size_t)(vw.q_row0 + q);
        if (nsplit <= 1) {
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            T* dst = O + row * p.ldo + head * 64 + fg * 4;
#pragma unroll
            for (int d = 0; d < 4; ++d) *reinterpret_cast<v4*>(dst + d * 16) = cvt4<T>(o_[d][f] * inv);
        } else if (p.part16) {   // normalised partial O_s / l_s in the 16-bit type: half the partial traffic
            const size_t D = (size_t)p.heads * 64;
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            T* po = reinterpret_cast<T*>(p.part_o) + ((size_t)split * p.total_q_rows + row) * D + head * 64 + fg * 4;
#pragma unroll
            for (int d = 0; d < 4; ++d) *reinterpret_cast<v4*>(po + d * 16) = cvt4<T>(o_[d][f] * inv);
            if (fg == 0) {
                float* pm = p.part_ml + (((size_t)split * p.total_q_rows + row) * p.heads + head) * 2;
                pm[0] = m_[f];
                pm[1] = l;
            }
        } else {
            const size_t D = (size_t)p.heads * 64;
            float* po = p.part_o + ((size_t)split * p.total_q_rows + row) * D + head * 64 + fg * 4;
#pragma unroll
            for (int d = 0; d < 4; ++d) *reinterpret_cast<f32x4*>(po + d * 16) = o_[d][f];
            if (fg == 0) {
                float* pm = p.part_ml + (((size_t)split * p.total_q_rows + row) * p.heads + head) * 2;
                pm[0] = m_[f];
                pm[1] = l;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// attn4_kernel (r03): the same transposed flash formulation on 32 x 32 MFMA tiles.
//   S^T tile = 32 keys x 32 queries: v_mfma_f32_32x32x16 (4 steps over the 64 head dims) or, with F8, ONE
//              v_mfma_scale_f32_32x32x64_f8f6f4 on e4m3 operands (MX scales = 1): twice the MFMA rate for Q K^T.
//   O^T tile = 32 head dims x 32 queries, k = 16 keys per v_mfma_f32_32x32x16 -- always 16-bit operands (P rounded to T, V in T):
//              a CPU emulation of the operand formats (scripts/emul/fp8_attention.py, MUSt3R_224, 8 views) puts e4m3 Q / K at
//              1.8e-3 of pointmap error, e4m3 P at 1.3e-3 and e4m3 V at 9e-3 -- V is the operand that cannot take 3 mantissa bits.
// Why 32 x 32: a 32-cycle MFMA leaves ~5 issue slots in its shadow, a 16-cycle one ~2 (MI355X_MICROARCH.md, per-instruction table);
// attn3_kernel is bound by the instructions it has to ISSUE beside its 36 short MFMAs per tile, this one issues 16 (8 + 8) long
// ones (F8: 2 + 8) for the same tile, and no `ones` MFMAs (the row sums are VALU adds in the MFMA shadows).
// Layouts (lane l: c = l & 31, h = l >> 5):
//   32x32x16   A[i][k]: i = c, k = 8 h + e (e < 8);   B[k][j]: j = c, k = 8 h + e;   C[i][j]: j = c, i = (r & 3) + 8 (r >> 2) + 4 h (r < 16)
//   32x32x64 (f8f6f4)   A / B: 32 bytes per lane, k = 32 h + e;   C as above
//   => lane (q = c, h) holds the scores of ONE query against keys {(r & 3) + 8 (r >> 2) + 4 h} of a 32-key tile; registers 8 m .. 8 m + 7
//      are keys {16 m + 4 h + (e & 3) + 8 (e >> 2)}: exactly a B fragment of the second product over the 16 keys [16 m, 16 m + 16),
//      and V^T comes from two transposing reads per fragment keyed the same way (keys base .. +3 and base + 8 .. +11, base = 16 m + 4 h).
//   Only the running maximum crosses lanes (l <-> l ^ 32, slow path only); the row sums are combined once, at the end.
// Memory / LDS: K tile [64 keys][64] T (F8: [64][64] bytes), V tile [64][64] T, staged by LDS-DMA exactly like attn3_kernel.
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// cbsz = blgp = 0: both operands OCP e4m3; scale bytes 127 = 2^0 (E8M0)
__device__ __forceinline__ f32x16 mfma32_mx8(i32x8 a, i32x8 b, f32x16 c) { return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127); }

// eight transposing reads = the V^T fragments of one 32-key half tile: (16-key step m = 0, 1) x (head-dim half dh = 0, 1) x (keys base, base + 8);
// a0 / a1 = the lane's address for dh = 0 / 1, O = byte offset of the half tile's first row (buffer, K|V, row) as an immediate
template <int O>
__device__ __forceinline__ void lds_tr_x8_issue(unsigned a0, unsigned a1, u32x2 (&r)[8]) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%10\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:%11\n\t"
        "ds_read_b64_tr_b16 %2, %9 offset:%10\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:%11\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %5, %8 offset:%13\n\t"
        "ds_read_b64_tr_b16 %6, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %7, %9 offset:%13"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "v"(a0), "v"(a1), "n"(O), "n"(O + 8 * 128), "n"(O + 16 * 128), "n"(O + 24 * 128)
        : "memory");
}
// the same, ordered BEFORE the consumers of an earlier fragment set: p[0..3] travel through the statement as read-write operands, so the MFMAs
// that take them (and, through their accumulators, the rest of that set's MFMAs) cannot be scheduled above these reads
template <int O>
__device__ __forceinline__ void lds_tr_x8_issue_after(unsigned a0, unsigned a1, u32x2 (&r)[8], u32x2 (&p)[8]) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %12 offset:%14\n\t"
        "ds_read_b64_tr_b16 %1, %12 offset:%15\n\t"
        "ds_read_b64_tr_b16 %2, %13 offset:%14\n\t"
        "ds_read_b64_tr_b16 %3, %13 offset:%15\n\t"
        "ds_read_b64_tr_b16 %4, %12 offset:%16\n\t"
        "ds_read_b64_tr_b16 %5, %12 offset:%17\n\t"
        "ds_read_b64_tr_b16 %6, %13 offset:%16\n\t"
        "ds_read_b64_tr_b16 %7, %13 offset:%17"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])
        : "v"(a0), "v"(a1), "n"(O), "n"(O + 8 * 128), "n"(O + 16 * 128), "n"(O + 24 * 128)
        : "memory");
}
// the same, ordered after the ARRIVAL of a 128-bit value X the compiler loaded itself (it must wait for X before this statement "modifies" it) and
// before X's consumer: issued from behind the last K-fragment read of a tile, these reads are not caught by the compiler's own lgkmcnt wait for it
template <int O, class X>
__device__ __forceinline__ void lds_tr_x8_issue_behind(unsigned a0, unsigned a1, u32x2 (&r)[8], X& x) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %9 offset:%11\n\t"
        "ds_read_b64_tr_b16 %1, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %2, %10 offset:%11\n\t"
        "ds_read_b64_tr_b16 %3, %10 offset:%12\n\t"
        "ds_read_b64_tr_b16 %4, %9 offset:%13\n\t"
        "ds_read_b64_tr_b16 %5, %9 offset:%14\n\t"
        "ds_read_b64_tr_b16 %6, %10 offset:%13\n\t"
        "ds_read_b64_tr_b16 %7, %10 offset:%14"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "+v"(x)
        : "v"(a0), "v"(a1), "n"(O), "n"(O + 8 * 128), "n"(O + 16 * 128), "n"(O + 24 * 128)
        : "memory");
}
// the reads above land asynchronously: every consumer takes the registers from THIS statement (read-write operands), so nothing
// the compiler schedules can use them before the counter says they have arrived.  LDS operations return in order, so the wait is
// also correct (conservative) for any compiler-issued ds_read in flight.
// (`late`: a value computed just before the consumers -- the first P fragment -- also travels through the statement, which keeps the
// scheduler from hoisting the wait to right behind the reads it waits for)
template <class X>
__device__ __forceinline__ void lds_tr_x8_wait(u32x2 (&r)[8], X& late) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(late)
                 :
                 : "memory");
}

__device__ __forceinline__ float max16x(const f32x16& a) {
    return max16f(__builtin_shufflevector(a, a, 0, 1, 2, 3), __builtin_shufflevector(a, a, 4, 5, 6, 7), __builtin_shufflevector(a, a, 8, 9, 10, 11),
                  __builtin_shufflevector(a, a, 12, 13, 14, 15));
}

template <class T, int QF, bool F8>   // QF: 32-query fragments per wave
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QF == 1 ? 2 : 1))) attn4_kernel(const AttnArgs p, const int nqb, const int ngrp, const int nsplit) {
    typedef typename Vec<T>::v8 v8;
    typedef typename Vec<T>::v4 v4;
    constexpr int QW = 32 * QF;
    constexpr int QB = 4 * QW;
    constexpr int KB = F8 ? ATT_KT * 64 : ATT_KT * 128;     // bytes of a K tile
    constexpr int VB = ATT_KT * 128;                        // bytes of a V tile
    constexpr int TB = KB + VB;
    __shared__ __attribute__((aligned(16))) char smem[2 * TB];   // [buffer][K tile | V tile]
    typedef __attribute__((address_space(3))) const char* lds_cptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 31, lh = lane >> 5;

    int grp, split, qb;
    if (!attn_block_coords(nqb, ngrp, nsplit, grp, split, qb)) return;
    const int view = grp / p.heads, head = grp - view * p.heads;
    const AttnView vw = p.view0_inline ? p.view0 : p.views[view];
    if (qb * QB >= vw.nq) return;

    const int kes = F8 ? 1 : 2;                              // bytes per Q / K element
    const char* __restrict__ Qb = reinterpret_cast<const char*>(p.Q);
    const char* __restrict__ Kb = reinterpret_cast<const char*>(p.K) + ((size_t)vw.kv_row0 * p.ldk + head * 64) * kes;
    const T* __restrict__ V = reinterpret_cast<const T*>(p.V) + (size_t)vw.kv_row0 * p.ldv + head * 64;

    // ---- Q fragments (B operand of the first product): lane (q = lc, h = lh)
    const int qr0 = qb * QB + wave * QW;
    v8 qf_[QF][F8 ? 1 : 4];
    i32x8 qf8_[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        int r = qr0 + f * 32 + lc;
        r = r < vw.nq ? r : vw.nq - 1;
        const char* src = Qb + ((size_t)(vw.q_row0 + r) * p.ldq + head * 64) * kes;
        if constexpr (F8) {
            const i32x4 lo = *reinterpret_cast<const i32x4*>(src + 32 * lh), hi = *reinterpret_cast<const i32x4*>(src + 32 * lh + 16);
            qf8_[f] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf_[f][ks] = *reinterpret_cast<const v8*>(src + (16 * ks + 8 * lh) * 2);
        }
    }

    const int nk = vw.nk, slo = vw.skip_lo, shi = vw.skip_hi;
    const int ntiles = (nk + ATT_KT - 1) / ATT_KT;
    auto fully_skipped = [&](int t) {
        const int k0 = t * ATT_KT;
        const int k1 = (k0 + ATT_KT < nk) ? k0 + ATT_KT : nk;
        return k0 >= slo && k1 <= shi;
    };
    const int tps = (ntiles + nsplit - 1) / nsplit;
    const int t_begin = split * tps;
    const int t_end = (t_begin + tps < ntiles) ? t_begin + tps : ntiles;
    auto advance = [&](int t) {
        ++t;
        if (shi > slo)
            while (t < t_end && fully_skipped(t)) ++t;
        return t;
    };

    // ---- staging through buffer descriptors: rows >= nk lie outside the descriptor and read as zero.  16-bit tiles: a wave
    // instruction moves 8 rows x 128 B (2 K + 2 V pieces per wave and tile); e4m3 K tile: 16 rows x 64 B (1 piece per wave).
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Kb), 0, ((nk - 1) * p.ldk + 64) * kes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(V), 0, ((nk - 1) * p.ldv + 64) * 2, 0x00020000);
    int vok[2], vov[2];
    {
        const int srow = lane >> 3, pch = lane & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (wave * 2 + i) * 8 + srow;
            vov[i] = (r * p.ldv + swz_v(r, pch) * 8) * 2;
            if constexpr (!F8) vok[i] = (r * p.ldk + swz(r, pch) * 8) * 2;
        }
        if constexpr (F8) {
            const int r = wave * 16 + (lane >> 2);
            vok[0] = r * p.ldk + swz32(r, lane & 3) * 16;
            vok[1] = 0;
        }
    }
    const int tstride_k = ATT_KT * p.ldk * kes, tstride_v = ATT_KT * p.ldv * 2;
    auto stage = [&](int t, int buf) {
        if constexpr (F8) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)&smem[buf * TB + wave * 16 * 64], 16, vok[0], t * tstride_k, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)&smem[buf * TB + (wave * 2 + i) * 8 * 128], 16, vok[i],
                                                         t * tstride_k, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)&smem[buf * TB + KB + (wave * 2 + i) * 8 * 128], 16, vov[i],
                                                     t * tstride_v, 0, 0);
    };

    // ---- per-lane LDS byte offsets
    //   K fragment, 16-bit (tile half t, step ks): row 32 t + lc, chunk (2 ks + lh) ^ ((lc >> 1) & 7) = (2 ks) ^ z
    //   K fragment, e4m3 (tile half t): row 32 t + lc, 16-byte chunks (2 lh) ^ x and (2 lh + 1) ^ x, x = swz32 pattern of the row
    //   V^T read (tile half t, step m, dim half dh, key group s): row 32 t + 16 m + 8 s + 4 lh' + e, lh' = lane >> 5, group g = lane >> 4,
    //     position pp = lane & 15 (e = pp >> 2, c = pp & 3), chunk pair (2 dh + (g & 1)) ^ ((rowlane >> 1) & 3), byte 8 c inside the pair
    // All of them are absolute LDS byte addresses inside buffer 0; after every tile they move by +TB / -TB.  (ONE loop body: the
    // loop-carried state -- 80+ registers of O, l, m, -m -- stays where it is; a body unrolled over the two buffers, with the buffer
    // in the immediates, made the register allocator copy the -m operand at every join.)
    const unsigned lds0 = (unsigned)(size_t)(lds_cptr)(&smem[0]);
    unsigned kaddr[F8 ? 2 : 4];
    if constexpr (!F8) {
        const int z = lh ^ ((lc >> 1) & 7);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kaddr[ks] = lds0 + (unsigned)(lc * 128 + ((2 * ks) ^ z) * 16);
    } else {
        kaddr[0] = lds0 + (unsigned)(lc * 64 + swz32(lc, 2 * lh) * 16);
        kaddr[1] = lds0 + (unsigned)(lc * 64 + swz32(lc, 2 * lh + 1) * 16);
    }
    unsigned vaddr[2];
    {
        const int g = lane >> 4, pp = lane & 15;
        const int rowlane = 4 * (g >> 1) + (pp >> 2);
        const int x = (rowlane >> 1) & 3;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) vaddr[dh] = lds0 + (unsigned)(KB + rowlane * 128 + (((2 * dh + (g & 1)) ^ x) * 32) + (pp & 3) * 8);
    }

    // ---- running state: o_[dh][f] = O^T rows 32 dh .. (16 registers), l_[f] = this lane's share of the row sum, m_[f] = reference of
    // query lc (same in both half waves), nm_[f] = -m/c splat: the C operand of the first MFMA of every score tile (S - m for free)
    f32x16 o_[2][QF], nm_[QF];
    float l_[QF], m_[QF];
    bool any_first = true;
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        m_[f] = -INFINITY;
        l_[f] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { nm_[f][r] = 0.f; o_[0][f][r] = 0.f; o_[1][f][r] = 0.f; }
    }
    const bool has_skip = shi > slo;
    const float c = p.q_prescaled ? 1.0f : p.scale * 1.44269504088896340736f;
    const float inv_c = 1.0f / c;

    auto compute = [&](int t) {
        // ---- S^T = K Q^T - m : two 32-key tiles
        f32x16 s_[2][QF];
        u32x2 tr[2][8];   // V^T fragments; those of the first 32 keys start their trip behind the last K fragment: they are needed after
                          // the maxima, the decision and the exponentials
        if constexpr (F8) {
            i32x4 kq[2][2];
#pragma unroll
            for (int th = 0; th < 2; ++th)
#pragma unroll
                for (int i = 0; i < 2; ++i) kq[th][i] = *reinterpret_cast<__attribute__((address_space(3))) const i32x4*>((size_t)(kaddr[i] + th * 32 * 64));
            lds_tr_x8_issue_behind<0>(vaddr[0], vaddr[1], tr[0], kq[1][1]);
#pragma unroll
            for (int th = 0; th < 2; ++th) {
                const i32x8 kfrag = __builtin_shufflevector(kq[th][0], kq[th][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int f = 0; f < QF; ++f) s_[th][f] = mfma32_mx8(kfrag, qf8_[f], nm_[f]);
            }
        } else {
            v8 kf[2][4];   // all eight fragment reads in flight before the first MFMA
#pragma unroll
            for (int th = 0; th < 2; ++th)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[th][ks] = *reinterpret_cast<__attribute__((address_space(3))) const v8*>((size_t)(kaddr[ks] + th * 32 * 128));
            lds_tr_x8_issue_behind<0>(vaddr[0], vaddr[1], tr[0], kf[1][3]);
#pragma unroll
            for (int th = 0; th < 2; ++th)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int f = 0; f < QF; ++f) s_[th][f] = mfma32(kf[th][ks], qf_[f][ks], ks == 0 ? nm_[f] : s_[th][f]);
        }
        if (!p.q_prescaled) {
#pragma unroll
            for (int th = 0; th < 2; ++th)
#pragma unroll
                for (int f = 0; f < QF; ++f) s_[th][f] *= c;
        }
        const int k0 = t * ATT_KT;
        const bool need_mask = (k0 + ATT_KT > nk) || (has_skip && k0 < shi && k0 + ATT_KT > slo);
        if (need_mask) {
            // branch-free bit arithmetic (rare path; 32 compare masks would cost 64 SGPRs and spill the loop's scalars):
            //   key >= nk                 <=>  (nk - 1 - key) < 0
            //   slo <= key < shi          <=>  ((key - slo) | (shi - 1 - key)) >= 0
            const int kb = k0 + 4 * lh;
#pragma unroll
            for (int th = 0; th < 2; ++th)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb + th * 32 + (r & 3) + 8 * (r >> 2);
                    const int past = (nk - 1 - key) >> 31;                          // all ones: beyond the last key
                    const int inside = ~(((key - slo) | (shi - 1 - key)) >> 31);   // all ones: inside the excluded range
                    const float pen = __int_as_float((past | inside) & (int)0xff800000u);   // -inf or +0
#pragma unroll
                    for (int f = 0; f < QF; ++f) s_[th][f][r] += pen;
                }
        }
        // ---- does any reference have to move?  one per-lane maximum over the lane's 32 scores of each query
        float mxa = -INFINITY;
        float mxl[QF];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            mxl[f] = fmaxf(max16x(s_[0][f]), max16x(s_[1][f]));
            mxa = f == 0 ? mxl[f] : fmaxf(mxa, mxl[f]);
        }
        if (any_first || __any(mxa > ATT_THR)) {   // wave-uniform, rare after the first tile: everything updated in place
            bool fst = false;
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                float mx = mxl[f];
                {   // the other half wave holds the other 32 keys of the same query
                    const unsigned u = __float_as_uint(mx);
                    auto q2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                    mx = fmaxf(__uint_as_float(q2[0]), __uint_as_float(q2[1]));
                }
                const bool first = (m_[f] == -INFINITY);
                float d = first ? mx : fmaxf(mx, 0.f);
                d = (d == -INFINITY) ? 0.f : d;                       // row still has no valid key
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-d);
                s_[0][f] -= d;
                s_[1][f] -= d;
                o_[0][f] *= alpha;
                o_[1][f] *= alpha;
                l_[f] *= alpha;
                m_[f] = first ? ((mx == -INFINITY) ? -INFINITY : d) : m_[f] + d;
                const float nm = (m_[f] == -INFINITY) ? 0.f : -m_[f] * inv_c;
#pragma unroll
                for (int r = 0; r < 16; ++r) nm_[f][r] = nm;
                fst |= (m_[f] == -INFINITY);
            }
            any_first = __any(fst);
        }
        // ---- O^T += V^T P^T, 16 keys per MFMA; the row sums ride along as VALU adds (four chains per query).  The transposing reads of
        // a 32-key half are issued one stage ahead of the MFMAs that consume them (the exponentials / the other half's MFMAs in between).
#pragma unroll
        for (int th = 0; th < 2; ++th)
#pragma unroll
            for (int f = 0; f < QF; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_[th][f][r] = __builtin_amdgcn_exp2f(s_[th][f][r]);
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            v8 pb[2][QF];
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                float a0 = s_[th][f][0] + s_[th][f][1], a1 = s_[th][f][2] + s_[th][f][3];
                float a2 = s_[th][f][4] + s_[th][f][5], a3 = s_[th][f][6] + s_[th][f][7];
#pragma unroll
                for (int r = 8; r < 16; r += 4) {
                    a0 += s_[th][f][r]; a1 += s_[th][f][r + 1]; a2 += s_[th][f][r + 2]; a3 += s_[th][f][r + 3];
                }
                l_[f] += (a0 + a1) + (a2 + a3);
                pb[0][f] = cvt8<T>(__builtin_shufflevector(s_[th][f], s_[th][f], 0, 1, 2, 3, 4, 5, 6, 7));
                pb[1][f] = cvt8<T>(__builtin_shufflevector(s_[th][f], s_[th][f], 8, 9, 10, 11, 12, 13, 14, 15));
            }
            lds_tr_x8_wait(tr[th], pb[0][0]);
            if (th == 0) lds_tr_x8_issue_after<32 * 128>(vaddr[0], vaddr[1], tr[1], tr[0]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    v4 lo, hi;
                    __builtin_memcpy(&lo, &tr[th][4 * m + 2 * dh], 8);
                    __builtin_memcpy(&hi, &tr[th][4 * m + 2 * dh + 1], 8);
                    const v8 vfrag = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int f = 0; f < QF; ++f) o_[dh][f] = mfma32(vfrag, pb[m][f], o_[dh][f]);
                }
        }
    };

    int t = advance(t_begin - 1);
    int buf = 0;
    if (t < t_end) stage(t, 0);
    while (t < t_end) {
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's pieces of tile t have landed
        __syncthreads();
        const int tn = advance(t);
        if (tn < t_end) stage(tn, buf ^ 1);
        compute(t);
        const unsigned step = buf ? (unsigned)(-TB) : (unsigned)TB;
#pragma unroll
        for (int i = 0; i < (F8 ? 2 : 4); ++i) kaddr[i] += step;
        vaddr[0] += step;
        vaddr[1] += step;
        buf ^= 1;
        t = tn;
    }

    // ---- normalise and store: lane (q = lc, h) holds O[q][d = 32 dh + 8 j + 4 h + (0..3)] in registers 4 j .. 4 j + 3 of o_[dh]
    T* __restrict__ O = reinterpret_cast<T*>(p.O);
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float l = l_[f];
        {
            const unsigned u = __float_as_uint(l);
            auto q2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            l = __uint_as_float(q2[0]) + __uint_as_float(q2[1]);
        }
        const int q = qr0 + f * 32 + lc;
        if (q >= vw.nq) continue;
        const size_t row = (size_t)(vw.q_row0 + q);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        if (nsplit <= 1 || p.part16) {   // final output, or normalised partial O_s / l_s in the 16-bit type
            T* dst = nsplit <= 1 ? O + row * p.ldo + head * 64 + lh * 4
                                 : reinterpret_cast<T*>(p.part_o) + ((size_t)split * p.total_q_rows + row) * ((size_t)p.heads * 64) + head * 64 + lh * 4;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = o_[dh][f][4 * j + r] * inv;
                    *reinterpret_cast<v4*>(dst + 32 * dh + 8 * j) = cvt4<T>(w);
                }
        } else {
            float* po = p.part_o + ((size_t)split * p.total_q_rows + row) * ((size_t)p.heads * 64) + head * 64 + lh * 4;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = o_[dh][f][4 * j + r];
                    *reinterpret_cast<f32x4*>(po + 32 * dh + 8 * j) = w;
                }
        }
        if (nsplit > 1 && lh == 0) {
            float* pm = p.part_ml + (((size_t)split * p.total_q_rows + row) * p.heads + head) * 2;
            pm[0] = m_[f];
            pm[1] = l;
        }
    }
}

// merge of the split-KV partials: O = sum_s 2^(m_s - m*) O_s / sum_s 2^(m_s - m*) l_s ; one thread = 4 columns
// TO_PARTIAL (r06, context-parallel cross attention): instead of the normalised 16-bit rows the merged sums themselves leave as ONE partial -- cp16 = 0: acc
// (un-normalised, fp32) to cp_o [rows][D]; cp16 = 1: acc / L in the 16-bit type (the split-KV partial format, half the bytes on the links) --, (m*, L) to cp_ml
// [rows][heads][2] -- which attn_combine_kernel<T, false> reads back (part16 = cp16) among the partials of the other ranks.
template <class T, bool TO_PARTIAL = false>
__global__ void attn_combine_kernel(const AttnArgs p, const int nsplit, float* __restrict__ cp_o = nullptr, float* __restrict__ cp_ml = nullptr, const int cp16 = 0) {
    typedef typename Vec<T>::v4 v4;
    const size_t D = (size_t)p.heads * 64;
    const size_t total = (size_t)p.total_q_rows * (D / 4);
    const size_t o_stride = p.part_stride_o > 0 ? (size_t)p.part_stride_o : (size_t)p.total_q_rows * D;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t row = idx / (D / 4);
        const int col = (int)(idx % (D / 4)) * 4;
        const int head = col >> 6;
        // All loads of a chunk of CH splits are issued before the first use: a loop that loads, waits and accumulates per split pays one
        // memory round trip per split (7-14 of them: ~8 us for a kernel that moves a few MB).  Same arithmetic, same order over the splits.
        constexpr int CH = 8;
        const size_t ml_stride = p.part_stride_ml > 0 ? (size_t)p.part_stride_ml : (size_t)p.total_q_rows * p.heads * 2;
        const float* ml0 = p.part_ml + (row * p.heads + head) * 2;
        float mstar = -INFINITY;
        for (int s0 = 0; s0 < nsplit; s0 += CH) {
            float mv[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int sc = s0 + c < nsplit ? s0 + c : nsplit - 1;
                mv[c] = ml0[(size_t)sc * ml_stride];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) mstar = fmaxf(mstar, mv[c]);
        }
        if (mstar == -INFINITY) {   // row not produced by any view of this launch (a partial: the row of a rank without keys)
            if constexpr (TO_PARTIAL) {
                if (cp16) *reinterpret_cast<v4*>(reinterpret_cast<T*>(cp_o) + row * D + col) = cvt4<T>(f32x4{0.f, 0.f, 0.f, 0.f});
                else *reinterpret_cast<f32x4*>(cp_o + row * D + col) = f32x4{0.f, 0.f, 0.f, 0.f};
                if ((col & 63) == 0) { cp_ml[(row * p.heads + head) * 2] = -INFINITY; cp_ml[(row * p.heads + head) * 2 + 1] = 0.f; }
            }
            continue;
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float L = 0.f;
        for (int s0 = 0; s0 < nsplit; s0 += CH) {
            float m_[CH], l_[CH];
            f32x4 o_[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int sc = s0 + c < nsplit ? s0 + c : nsplit - 1;
                const float* ml = ml0 + (size_t)sc * ml_stride;
                m_[c] = ml[0];
                l_[c] = ml[1];
                if (p.part16) o_[c] = __builtin_convertvector(*reinterpret_cast<const v4*>(reinterpret_cast<const T*>(p.part_o) + (size_t)sc * o_stride + row * D + col), f32x4);
                else o_[c] = *reinterpret_cast<const f32x4*>(p.part_o + (size_t)sc * o_stride + row * D + col);
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (s0 + c >= nsplit) break;
                float w = __builtin_amdgcn_exp2f(m_[c] - mstar);
                L += w * l_[c];
                if (p.part16) w *= l_[c];   // partial holds O_s / l_s
                acc += o_[c] * w;
            }
        }
        if constexpr (TO_PARTIAL) {
            if (cp16) *reinterpret_cast<v4*>(reinterpret_cast<T*>(cp_o) + row * D + col) = cvt4<T>(acc * (L > 0.f ? 1.0f / L : 0.f));
            else *reinterpret_cast<f32x4*>(cp_o + row * D + col) = acc;
            if ((col & 63) == 0) { cp_ml[(row * p.heads + head) * 2] = mstar; cp_ml[(row * p.heads + head) * 2 + 1] = L; }
        } else {
            const float inv = L > 0.f ? 1.0f / L : 0.f;
            *reinterpret_cast<v4*>(reinterpret_cast<T*>(p.O) + row * p.ldo + col) = cvt4<T>(acc * inv);
        }
    }
}

// fills (m, l) = (-inf, 0) so rows no block writes are recognisable by the combine pass
__global__ void attn_ml_init_kernel(float* ml, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        ml[i * 2] = -INFINITY;
        ml[i * 2 + 1] = 0.f;
    }
}

size_t attention_split_scratch_bytes(int nsplit, int total_q_rows, int heads) {
    if (nsplit <= 1) return 0;
    return (size_t)nsplit * total_q_rows * ((size_t)heads * 64 * 4 + (size_t)heads * 8) + 512;
}

bool attention_is_small(int nviews, int heads, int max_nq, int nsplit) {
    return nsplit <= 1 && (long)nviews * heads * ((max_nq + ATT_QB - 1) / ATT_QB) < 192;
}

int attention_pick_split(int nviews, int heads, int max_nq, int max_nk) {
    // Measured model (r02, scripts/bench_attn.py + scripts/probes/attn_trace.hip): a block needs ~1.4 us per 64-key tile whether
    // two or three blocks share its CU (the wave's own QK -> softmax -> PV chain, not the CU's throughput, sets the pace), three
    // blocks fit a CU (163 VGPRs), and a split launch pays ~26 us of fixed cost (prologue, fp32 partials, combine launch) plus the
    // partial traffic that grows with the factor.  So: as many blocks as the 768 slots take in ONE round, i.e. as few tiles per
    // block as possible, never a second round.  One view x 12 heads x 6 query blocks: s = 10 (720 blocks; nk = 7680: 43 us against
    // 52 us at s = 7 and 47 us at s = 14).
    const long base = (long)nviews * heads * ((max_nq + ATT_QB - 1) / ATT_QB);
    const int ntiles = (max_nk + ATT_KT - 1) / ATT_KT;
    if (base >= 384 || ntiles < 8) return 1;
    int best = 1;
    double best_cost = 1e30;
    const int smax = ntiles / 4 < 16 ? ntiles / 4 : 16;   // at least 4 key tiles per block
    for (int s = 1; s <= smax; ++s) {
        const long blocks = base * s;
        const long rounds = (blocks + 767) / 768;
        const int tiles = (ntiles + s - 1) / s;
        const double cost = (double)rounds * tiles + (s > 1 ? 3.0 + 0.5 * s : 0.0);   // in tile units (~1.4 us)
        if (cost < best_cost) { best_cost = cost; best = s; }
    }
    return best;
}

// hardware-semantics probe used by the tests: LDS holds element index e at position e; every lane issues the
// transposing read on its canonical chunk (lane*4 elements) and reports the 4 values it received.
__global__ void tr_probe_kernel(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[256];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 4));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = r[e];
}
int launch_tr_probe(short* out, hipStream_t s) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, s, out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// which main kernel the last launch_attention_phase(.., 1, ..) of this thread ran (profile rows are keyed by it: one rocprofv3 symbol each)
static thread_local const char* g_attn_pick = "";
const char* attention_last_kernel() { return g_attn_pick; }

// phase 0: (m,l) pre-fill (split-KV with holes only), 1: main kernel, 2: combine (split-KV only)
int launch_attention_phase(DType dt, const AttnArgs& a_in, int phase, hipStream_t s, const char** err) {
    AttnArgs a = a_in;
    if (a.nviews <= 0 || a.max_nq <= 0) return 0;
    {   // split-KV partials of the fp16 paths are written normalised, in fp16 (bf16 keeps fp32)
        a.part16 = (dt == DT_F16 && a.nsplit > 1) ? 1 : 0;
    }
    if ((a.ldq % (a.fp8 ? 16 : 8)) || (a.ldk % (a.fp8 ? 16 : 8)) || (a.ldv % (a.fp8 ? 16 : 8)) || (a.ldo % 4)) { *err = "attention: row strides must be 16-byte aligned"; return 1; }
    const int nsplit = a.nsplit > 1 ? a.nsplit : 1;
    if (nsplit > 1 && (!a.part_o || !a.part_ml || a.total_q_rows <= 0)) { *err = "attention: split-KV needs scratch"; return 1; }
    // Launches that cannot fill 256 CUs even with one block each (a single view's self attention in the memory
    // update: 12 heads x 6 query blocks) run with 16 query rows per wave: twice the blocks, half the serial work each.
    const int ngrp = a.nviews * a.heads;
    // (measured: for the split-KV cross attention the 16-row variant is slower, 24.0 vs 22.7 ms per scene)
    // (fp8 operands only exist for the 32 x 32 kernel, which has no 16-row form: the model keeps such launches on 16-bit operands)
    const bool small = !a.fp8 && attention_is_small(a.nviews, a.heads, a.max_nq, nsplit);
    // 32-bit byte offsets inside one view's K / V rows (buffer descriptors, per-lane voffsets): refuse what would wrap
    if (a.max_nk > 0 && ((long long)a.max_nk * a.ldk * (a.fp8 ? 1 : 2) >= 0x7fffffffLL || (long long)a.max_nk * a.ldv * 2 >= 0x7fffffffLL)) {
        *err = "attention: a view's K / V rows span 2 GiB or more (32-bit staging offsets)";
        return 1;
    }
    // Experiment builds only (make EXTRA=-DM3R_ATTN_EXPERIMENTS): M3R_ATTN_QW = 48 / 64 query rows per wave for the launches that are not
    // `small` (equal / spilling, profiles/r02_attn3_ablation.txt) and the M3R_ATTN_ABL timing ablations of attn3_kernel, which compute
    // WRONG results on purpose -- neither is compiled into the product library.
#ifdef M3R_ATTN_EXPERIMENTS
    static int qw_big = -1;
    if (qw_big < 0) {
        const char* e = getenv("M3R_ATTN_QW");
        qw_big = e ? atoi(e) : 32;
    }
#else
    constexpr int qw_big = 32;
#endif
#ifdef M3R_ATTN_EXPERIMENTS
    // M3R_ATTN=4 M3R_ATTN_QF=2: 64 query rows per wave in the 32 x 32 kernel (one wave per SIMD, 512 registers)
    static const bool qf2_env = getenv("M3R_ATTN_QF") && atoi(getenv("M3R_ATTN_QF")) == 2 && getenv("M3R_ATTN") && atoi(getenv("M3R_ATTN")) == 4;
    const bool qf2 = qf2_env && !small && !a.fp8;
#else
    constexpr bool qf2 = false;
#endif
    const int qb_rows = small ? 64 : (qf2 ? 256 : 4 * qw_big);
    const int nqb_abs = (a.max_nq + qb_rows - 1) / qb_rows;
    const int npairs = ngrp * nsplit;
    // pairs dealt over the 8 XCDs; per-block round robin when that would leave the XCDs more than 10 % apart (see attn_block_coords)
    const bool per_block = ((npairs + 7) / 8) * 8 * 10 > npairs * 11;
    const int nqb = per_block ? -nqb_abs : nqb_abs;
    const int grid = per_block ? npairs * nqb_abs : ((npairs + 7) / 8) * 8 * nqb_abs;
    if (phase == 0) {
        if (nsplit > 1 && !a.dense_rows) {
            const size_t n2 = (size_t)nsplit * a.total_q_rows * a.heads;
            hipLaunchKernelGGL(attn_ml_init_kernel, dim3((unsigned)((n2 + 255) / 256 < 1024 ? (n2 + 255) / 256 : 1024)), dim3(256), 0, s,
                               a.part_ml, n2);
        }
    } else if (phase == 1) {
        // 16-bit operands: attn3_kernel (16 x 16 tiles).  The 32 x 32-tile attn4_kernel measures EQUAL on every shape of the scene (r03:
        // render cross attention 840 vs 852 TF/s, encoder self attention 699 vs 710; 64 query rows per wave at one wave per SIMD: 497 --
        // profiles/r03_attn_ab.txt), so the 16 x 16 kernel -- whose 16-row form also serves the launches too small to fill the chip, with
        // the same per-query arithmetic, i.e. bit-identical batched and per-view calls -- stays the 16-bit path, and attn4 is the
        // fp8 (MX-scaled Q K^T) path: 983-1003 TF/s on the render shape.  M3R_ATTN=4 (experiment builds) runs its 16-bit instantiation.
#define M3R_LAUNCH_ATTN(KERNEL) hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(256), 0, s, a, nqb, ngrp, nsplit)
        g_attn_pick = a.fp8 ? "attn4f8" : (small ? "attn3/q16" : "attn3/q32");
        if (a.fp8) {   // e4m3 Q / K through the MX-scaled 32x32x64 MFMA, 16-bit P / V -- parked (kernels.hpp kAttnFp8Built): make EXTRA=-DM3R_ATTN_FP8
#ifdef M3R_ATTN_FP8
            if (dt == DT_BF16) M3R_LAUNCH_ATTN((attn4_kernel<bf16_t, 1, true>)); else M3R_LAUNCH_ATTN((attn4_kernel<f16_t, 1, true>));
#else
            *err = "attention: e4m3 operands are an experiment build (make EXTRA=-DM3R_ATTN_FP8)";
            return 1;
#endif
        } else {
#ifdef M3R_ATTN_EXPERIMENTS
            static const int variant = getenv("M3R_ATTN") ? atoi(getenv("M3R_ATTN")) : 2;
            static const int abl = getenv("M3R_ATTN_ABL") ? atoi(getenv("M3R_ATTN_ABL")) : 0;   // timing ablations (wrong results)
            if (variant == 4 && !small) {
                if (qf2) { if (dt == DT_BF16) M3R_LAUNCH_ATTN((attn4_kernel<bf16_t, 2, false>)); else M3R_LAUNCH_ATTN((attn4_kernel<f16_t, 2, false>)); }
                else if (dt == DT_BF16) M3R_LAUNCH_ATTN((attn4_kernel<bf16_t, 1, false>)); else M3R_LAUNCH_ATTN((attn4_kernel<f16_t, 1, false>));
            }
            else if (!small && dt == DT_F16 && abl == 1) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 1>));
            else if (!small && dt == DT_F16 && abl == 2) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 2>));
            else if (!small && dt == DT_F16 && abl == 3) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 3>));
            else if (!small && dt == DT_F16 && abl == 4) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 4>));
            else if (!small && dt == DT_F16 && abl == 5) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 5>));
            else if (!small && dt == DT_F16 && abl == 6) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 6>));
            else if (!small && dt == DT_F16 && abl == 7) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 7>));
            else if (!small && dt == DT_F16 && abl == 8) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 8>));
            else if (!small && dt == DT_F16 && abl == 9) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 9>));
            else if (!small && dt == DT_F16 && qw_big == 48) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 48>));
            else if (!small && dt == DT_F16 && qw_big == 64) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 64>));
            else
#endif
            {
                // M3R_ATTN_LZ (A/B instrument): 1 (default) = references move on the tile's row sums, 0 = on the per-lane score maxima (r02-r04)
                const int lz = opt(OPT_ATTN_LZ);
#ifdef M3R_ATTN_EXPERIMENTS
                // measured and not kept (profiles/r05_attn_step_ab.txt): M3R_ATTN_NB=3 K/V tiles staged two tiles ahead through three buffers (-1.9 % on the step),
                // M3R_ATTN_PR=1 / 2 s_setprio 1 around the MFMA clusters / the exp2 cluster (+3 % on random operands, 0 / -0.9 % on the step)
                static const int nb = getenv("M3R_ATTN_NB") ? atoi(getenv("M3R_ATTN_NB")) : 2;
                static const int pr = getenv("M3R_ATTN_PR") ? atoi(getenv("M3R_ATTN_PR")) : 0;
                if (lz && pr == 1 && !small && dt == DT_F16) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 0, 1, 2, 1>));
                else if (lz && pr == 2 && !small && dt == DT_F16) M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 0, 1, 2, 2>));
                else if (lz && nb == 3) {
                    if (small) { if (dt == DT_BF16) M3R_LAUNCH_ATTN((attn3_kernel<bf16_t, 16, 0, 1, 3>)); else M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 16, 0, 1, 3>)); }
                    else { if (dt == DT_BF16) M3R_LAUNCH_ATTN((attn3_kernel<bf16_t, 32, 0, 1, 3>)); else M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 0, 1, 3>)); }
                } else if (lz == 2 && !small) {   // M3R_ATTN_LZ=2: the two 16-query halves of a wave one stage apart (32-row form only): -3.5 % on the render launch, -1 % on the step
                    if (dt == DT_BF16) M3R_LAUNCH_ATTN((attn3_kernel<bf16_t, 32, 0, 2>)); else M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 0, 2>));
                } else
#endif
                if (lz) {
                    if (small) { if (dt == DT_BF16) M3R_LAUNCH_ATTN((attn3_kernel<bf16_t, 16, 0, 1>)); else M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 16, 0, 1>)); }
                    else { if (dt == DT_BF16) M3R_LAUNCH_ATTN((attn3_kernel<bf16_t, 32, 0, 1>)); else M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32, 0, 1>)); }
                } else {
                    if (small) { if (dt == DT_BF16) M3R_LAUNCH_ATTN((attn3_kernel<bf16_t, 16>)); else M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 16>)); }
                    else { if (dt == DT_BF16) M3R_LAUNCH_ATTN((attn3_kernel<bf16_t, 32>)); else M3R_LAUNCH_ATTN((attn3_kernel<f16_t, 32>)); }
                }
            }
        }
#undef M3R_LAUNCH_ATTN
    } else if (nsplit > 1) {
        const size_t total = (size_t)a.total_q_rows * a.heads * 16;
        const unsigned g2 = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        if (dt == DT_BF16) hipLaunchKernelGGL((attn_combine_kernel<bf16_t, false>), dim3(g2), dim3(256), 0, s, a, nsplit, (float*)nullptr, (float*)nullptr, 0);
        else hipLaunchKernelGGL((attn_combine_kernel<f16_t, false>), dim3(g2), dim3(256), 0, s, a, nsplit, (float*)nullptr, (float*)nullptr, 0);
    }
    if (hipGetLastError() != hipSuccess) { *err = "attention: kernel launch failed"; return 1; }
    return 0;
}

// ---- context-parallel cross attention (kernels.hpp): partial merge / empty partial / final merge over the ranks' slots
static unsigned combine_grid(const AttnArgs& a) {
    const size_t total = (size_t)a.total_q_rows * a.heads * 16;
    return (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
}
int launch_attention_partial_merge(DType dt, const AttnArgs& a_in, float* out_o, float* out_ml, int p16, hipStream_t s, const char** err) {
    AttnArgs a = a_in;
    if (a.nsplit < 2 || !a.part_o || !a.part_ml || a.total_q_rows <= 0 || !out_o || !out_ml) { *err = "attention: partial merge needs split-KV partials and an output slot"; return 1; }
    a.part16 = dt == DT_F16 ? 1 : 0;   // the layout launch_attention_phase(.., 1, ..) wrote
    a.part_stride_o = a.part_stride_ml = 0;
    if (dt == DT_BF16) hipLaunchKernelGGL((attn_combine_kernel<bf16_t, true>), dim3(combine_grid(a)), dim3(256), 0, s, a, a.nsplit, out_o, out_ml, p16);
    else hipLaunchKernelGGL((attn_combine_kernel<f16_t, true>), dim3(combine_grid(a)), dim3(256), 0, s, a, a.nsplit, out_o, out_ml, p16);
    if (hipGetLastError() != hipSuccess) { *err = "attention: kernel launch failed"; return 1; }
    return 0;
}
__global__ void attn_partial_empty_kernel(float* __restrict__ o, float* __restrict__ ml, size_t n_o4, size_t n_ml) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n_o4 > n_ml ? n_o4 : n_ml); i += (size_t)gridDim.x * blockDim.x) {
        if (i < n_o4) reinterpret_cast<f32x4*>(o)[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // (zero bits are zero in fp32 and in both 16-bit types)
        if (i < n_ml) { ml[i * 2] = -INFINITY; ml[i * 2 + 1] = 0.f; }
    }
}
int launch_attention_partial_empty(float* out_o, float* out_ml, int rows, int heads, int p16, hipStream_t s, const char** err) {
    if (rows <= 0) return 0;
    const size_t n_o4 = (size_t)rows * heads * (p16 ? 8 : 16), n_ml = (size_t)rows * heads;
    const size_t n = n_o4 > n_ml ? n_o4 : n_ml;
    hipLaunchKernelGGL(attn_partial_empty_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, out_o, out_ml, n_o4, n_ml);
    if (hipGetLastError() != hipSuccess) { *err = "attention: kernel launch failed"; return 1; }
    return 0;
}
int launch_attention_partial_final(DType dt, const AttnArgs& a_in, const float* slots_o, const float* slots_ml, long long stride_o, long long stride_ml, int nslots,
                                   int p16, hipStream_t s, const char** err) {
    AttnArgs a = a_in;
    if (nslots < 1 || !slots_o || !slots_ml || a.total_q_rows <= 0 || !a.O || (a.ldo % 4)) { *err = "attention: final merge needs slots and an output"; return 1; }
    a.part_o = const_cast<float*>(slots_o);
    a.part_ml = const_cast<float*>(slots_ml);
    a.part16 = p16 ? 1 : 0;
    a.part_stride_o = stride_o;
    a.part_stride_ml = stride_ml;
    if (dt == DT_BF16) hipLaunchKernelGGL((attn_combine_kernel<bf16_t, false>), dim3(combine_grid(a)), dim3(256), 0, s, a, nslots, (float*)nullptr, (float*)nullptr, 0);
    else hipLaunchKernelGGL((attn_combine_kernel<f16_t, false>), dim3(combine_grid(a)), dim3(256), 0, s, a, nslots, (float*)nullptr, (float*)nullptr, 0);
    if (hipGetLastError() != hipSuccess) { *err = "attention: kernel launch failed"; return 1; }
    return 0;
}

int launch_attention(DType dt, const AttnArgs& a, hipStream_t s, const char** err) {
    for (int ph = 0; ph < 3; ++ph)
        if (launch_attention_phase(dt, a, ph, s, err)) return 1;
    return 0;
}

}  // namespace m3r
