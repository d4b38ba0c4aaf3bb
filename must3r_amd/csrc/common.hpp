// Shared device helpers for the gfx950 (CDNA4) kernels of the MUSt3R multi-view forward path.
// Wave = 64 lanes. MFMA shape used everywhere: v_mfma_f32_16x16x32_{bf16,f16}
//   A[i][k]: lane l holds i = l&15, k = (l>>4)*8 .. +7     (8 x 16-bit, 16 bytes)
//   B[k][j]: lane l holds j = l&15, k = (l>>4)*8 .. +7
//   C[i][j]: lane l holds j = l&15, i = (l>>4)*4 + r, r = 0..3  (f32x4)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace m3r {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <class T> struct Vec;
template <> struct Vec<bf16_t> { typedef bf16x8 v8; typedef bf16x4 v4; };
template <> struct Vec<f16_t>  { typedef f16x8 v8;  typedef f16x4 v4; };

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <class T> __device__ __forceinline__ typename Vec<T>::v4 cvt4(f32x4 v) {
    return __builtin_convertvector(v, typename Vec<T>::v4);
}
// saturating form for the GEMM epilogues: an fp16 store of |v| > 65504 would be +-inf and turn into NaN downstream (LayerNorm,
// softmax); the value is clamped to the largest finite fp16 instead (in-range values: same bits).  bf16 has fp32's range.
template <class T> __device__ __forceinline__ typename Vec<T>::v4 cvt4_sat(f32x4 v) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, bf16_t)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], -65504.0f, 65504.0f);
    }
    return __builtin_convertvector(v, typename Vec<T>::v4);
}
template <class T> __device__ __forceinline__ typename Vec<T>::v8 cvt8(f32x8 v) {
    return __builtin_convertvector(v, typename Vec<T>::v8);
}

// LDS transposed read (ds_read_b64_tr_b16): each lane passes the address of its own 8-byte (4 x 16-bit) chunk; inside a
// 16-lane group the 16 chunks form a 4 x 16 row-major matrix X (chunk m = row m>>2, cols 4*(m&3)..+3) and lane i of the
// group receives column i: {X[0][i], X[1][i], X[2][i], X[3][i]}  (checked on hardware: must3r_hip_debug_tr_probe).
// Eight transposing reads (one 32-key slot of V^T: 4 d-fragments x {keys 0-15, keys 16-31}) issued from inline asm
// together with their own lgkmcnt(0).  hipcc orders the builtin form behind every in-flight LDS-DMA (it emits
// s_waitcnt vmcnt(0) before it), which would drain the next tile's prefetch in the middle of the current tile; the
// asm form is invisible to that bookkeeping.  The reads only touch the tile buffer whose DMA was already waited for.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <class T>
__device__ __forceinline__ void lds_read_tr4_x8(const T* p0, const T* p1, const T* p2, const T* p3, const T* p4, const T* p5,
                                                const T* p6, const T* p7, typename Vec<T>::v4 (&out)[8]) {
    u32x2 r0, r1, r2, r3, r4, r5, r6, r7;
#define M3R_LDS_ADDR(p) ((unsigned)(size_t)(__attribute__((address_space(3))) const T*)(p))
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\t"
        "ds_read_b64_tr_b16 %1, %9\n\t"
        "ds_read_b64_tr_b16 %2, %10\n\t"
        "ds_read_b64_tr_b16 %3, %11\n\t"
        "ds_read_b64_tr_b16 %4, %12\n\t"
        "ds_read_b64_tr_b16 %5, %13\n\t"
        "ds_read_b64_tr_b16 %6, %14\n\t"
        "ds_read_b64_tr_b16 %7, %15\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
        : "v"(M3R_LDS_ADDR(p0)), "v"(M3R_LDS_ADDR(p1)), "v"(M3R_LDS_ADDR(p2)), "v"(M3R_LDS_ADDR(p3)), "v"(M3R_LDS_ADDR(p4)),
          "v"(M3R_LDS_ADDR(p5)), "v"(M3R_LDS_ADDR(p6)), "v"(M3R_LDS_ADDR(p7))
        : "memory");
#undef M3R_LDS_ADDR
    const u32x2 r[8] = {r0, r1, r2, r3, r4, r5, r6, r7};
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_memcpy(&out[i], &r[i], 8);
}

// four transposing reads (two d-fragments of one 32-key slot) -- same contract as lds_read_tr4_x8, half the live registers
template <class T>
__device__ __forceinline__ void lds_read_tr4_x4(const T* p0, const T* p1, const T* p2, const T* p3, typename Vec<T>::v4 (&out)[4]) {
    u32x2 r0, r1, r2, r3;
#define M3R_LDS_ADDR(p) ((unsigned)(size_t)(__attribute__((address_space(3))) const T*)(p))
    asm volatile(
        "ds_read_b64_tr_b16 %0, %4\n\t"
        "ds_read_b64_tr_b16 %1, %5\n\t"
        "ds_read_b64_tr_b16 %2, %6\n\t"
        "ds_read_b64_tr_b16 %3, %7\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(M3R_LDS_ADDR(p0)), "v"(M3R_LDS_ADDR(p1)), "v"(M3R_LDS_ADDR(p2)), "v"(M3R_LDS_ADDR(p3))
        : "memory");
#undef M3R_LDS_ADDR
    const u32x2 r[4] = {r0, r1, r2, r3};
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_memcpy(&out[i], &r[i], 8);
}

// async global -> LDS, 16 bytes per lane; LDS destination = (wave-uniform) lds_base + lane*16.
__device__ __forceinline__ void glds16(const void* gptr, void* lds_base_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_base_uniform, 16, 0, 0);
}

// max over the 4 lanes {l, l^16, l^32, l^48} with the gfx950 VALU half/row swaps (no LDS round trip, unlike __shfl_xor =
// ds_bpermute).  v_permlane32_swap(a, b): a' = {a.lo32, b.lo32}, b' = {a.hi32, b.hi32}; v_permlane16_swap swaps the odd
// 16-lane rows of a with the even rows of b.  Called with a = b = x both results together hold x[l] and x[l ^ 32 / 16].
__device__ __forceinline__ float quad_row_max(float x) {
    unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    u = __float_as_uint(x);
    auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

// sum over the 64 lanes without LDS round trips: four DPP row rotations (every lane of a 16-lane row ends with the row sum), then the
// two cross-row swaps of quad_row_max.  ~8 VALU instructions against six ds_bpermute round trips (~100+ cycles each) of the
// __shfl_xor form below; the one-row-per-wave LayerNorm of the memory update is latency-bound, so this is wall time there.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_mov<0x128>(v);   // row_ror:8
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x122>(v);   // row_ror:2
    v += dpp_mov<0x121>(v);   // row_ror:1
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    u = __float_as_uint(v);
    auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// sum over the 4 lanes {l, l^16, l^32, l^48} (the four column groups of one output row in the GEMM epilogues)
__device__ __forceinline__ float quad_row_sum(float x) {
    unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    u = __float_as_uint(x);
    auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 16-byte-chunk XOR swizzle for [rows][64 x 16-bit] (128-byte rows) LDS tiles read with ds_read_b128:
// physical chunk = logical chunk ^ ((row >> 1) & 7).  Rows r, r+1 sit in different halves of the 256-byte
// bank row, so 16 consecutive rows reading the same logical chunk hit 16 distinct 16-byte slots.
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
// The same for [rows][32 x 16-bit] (64-byte rows, 4 chunks): 4 rows share a 256-byte bank row.  The XOR pattern per
// row quad {0,2,3,1} keeps the hardware's ds_read_b128 service groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...)
// conflict-free when lanes 0-15 / 16-31 read chunk c / c^1 of rows r0..r0+15.
__device__ __forceinline__ int swz32(int row, int chunk) {
    const int q = (row >> 2) & 3;
    return chunk ^ ((0x1320 >> (q * 4)) & 3);   // q: 0,1,2,3 -> 0,2,3,1
}
// [rows][128 x 16-bit] (256-byte rows = one full bank row, 16 chunks): every row puts logical chunk c on the same banks, so the 16 rows of a
// fragment read would collide 16 ways; XOR with the row's low 4 bits spreads them over the 16 slots (and keeps the two halves of a
// ds_read_b128 service group -- rows {0-3, 12-15} at chunk c, rows {4-11} at chunk c ^ 1 -- on disjoint slots).
__device__ __forceinline__ int swz128(int row, int chunk) { return chunk ^ (row & 15); }
template <int BK> __device__ __forceinline__ int swzk(int row, int chunk) {
    if constexpr (BK == 128) return swz128(row, chunk);
    else if constexpr (BK == 64) return swz(row, chunk);
    else return swz32(row, chunk);
}

// V tiles are consumed by ds_read_b64_tr_b16: a 32-lane half reads 8 consecutive rows x one aligned 32-byte pair of
// chunks.  XOR on the PAIR index (bits 1-2 of the chunk) with (row>>1)&3 puts the 8 rows on 8 distinct 32-byte slots
// of the 256-byte bank row (rows r, r+1 already sit in different halves); swz() would alias rows r and r+2.
__device__ __forceinline__ int swz_v(int row, int chunk) { return chunk ^ (((row >> 1) & 3) << 1); }

// exact-erf GELU (nn.GELU default), r04 form:  GELU(x) = max(x, 0) - (|x| / 2) erfc(|x| / sqrt 2)  with  erfc(z) ~ 2^(-z P(z)),
// P a quartic fitted to -log2(erfc(z)) / z on [0, 4.3] (scripts/emul: least-squares start, reweighted to the minimax of the ABSOLUTE erfc error):
// |erfc error| <= 5.9e-7, |GELU error| <= 1.3e-6 absolute in fp32 arithmetic (relative 6e-5 where |GELU| > 1e-2) -- an eighth of the 16-bit rounding
// of the result.  P is positive and z P(z) increasing for every z >= 0 (leading coefficient > 0), so large |x| need no clamp: 2^(-z P) underflows to 0
// and the result is max(x, 0) exactly; the negative tail is a product, not a cancellation (0.5 x (1 + erf) loses the tail to 1 - 1).
// ONE transcendental (exp2) + 6 FMAs per value instead of two (rcp, exp2) + 7 FMAs + 6 other of the Abramowitz-Stegun 7.1.26 form used through r03 (1.5e-7): the erf
// epilogue was ~8 us of a 39 us fc1 tile, VALU-bound with the quarter-rate transcendentals at 44 % of it.
// Coefficients in a = |x| directly (Q_k = P_k / sqrt(2)^(k+1)), and the factor 1/2 folded into the exponent:
//   GELU(x) = max(x, 0) - a 2^(-a Q(a) - 1)                                   4 + 1 + 1 FMAs, one exp2
constexpr float kErfcQ0 = 1.1510913372039795f, kErfcQ1 = 0.4592546820640564f, kErfcQ2 = 0.052561257034540176f, kErfcQ3 = -0.007397521752864122f,
                kErfcQ4 = 0.0005204606568440795f;
__device__ __forceinline__ float gelu_erf(float x) {
    const float a = fabsf(x);
    float q = fmaf(kErfcQ4, a, kErfcQ3);
    q = fmaf(q, a, kErfcQ2);
    q = fmaf(q, a, kErfcQ1);
    q = fmaf(q, a, kErfcQ0);
    const float e = __builtin_amdgcn_exp2f(fmaf(-a, q, -1.0f));   // erfc(|x| / sqrt 2) / 2
    return fmaf(-a, e, fmaxf(x, 0.0f));
}

// The same arithmetic on two values at once: clang maps the ext_vector float2 operations to the packed fp32 VALU instructions of gfx950
// (v_pk_fma_f32 -- IEEE per component, i.e. the bits of the scalar form).  (Packed fp32 beside MFMAs is an anti-lever; in a GEMM epilogue no MFMA
// runs on the CU.)
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(const f32x2 x) {
    f32x2 a, r;
    a[0] = fabsf(x[0]);
    a[1] = fabsf(x[1]);
    r[0] = fmaxf(x[0], 0.0f);
    r[1] = fmaxf(x[1], 0.0f);
    f32x2 q = __builtin_elementwise_fma(f32x2{kErfcQ4, kErfcQ4}, a, f32x2{kErfcQ3, kErfcQ3});
    q = __builtin_elementwise_fma(q, a, f32x2{kErfcQ2, kErfcQ2});
    q = __builtin_elementwise_fma(q, a, f32x2{kErfcQ1, kErfcQ1});
    q = __builtin_elementwise_fma(q, a, f32x2{kErfcQ0, kErfcQ0});
    const f32x2 t = __builtin_elementwise_fma(-a, q, f32x2{-1.0f, -1.0f});
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(t[0]);
    e[1] = __builtin_amdgcn_exp2f(t[1]);
    return __builtin_elementwise_fma(-a, e, r);
}
__device__ __forceinline__ f32x4 gelu_erf4(const f32x4 v) {
    const f32x2 lo = gelu_erf2(f32x2{v[0], v[1]}), hi = gelu_erf2(f32x2{v[2], v[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

}  // namespace m3r
