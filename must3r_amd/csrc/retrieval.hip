// the retrieval front-end on the encoder tokens (must3r/retrieval/model.py:59-101,
// 165-183): Whitener (centre + PCA projection in float64, :67-79), projector Linear (:139-151), token attention = L2 norm
// (:130-131), top-k local feature selection (how_select_local, :91-101) and weighted sum pooling (weighted_spoc, :82-88).
//
//   gemmx_kernel<TC>     out[M,N] = fp32( (A[M,K] - sub[K]) . B + bias + resid ),  TC = double (Whitener: the reference
//                        computes it in float64) or float (projector); 64 x 64 x 16 tiles through LDS, 4 waves,
//                        v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 (exact products, TC accumulation).
//                        B is [K,N] row-major (Whitener.p) or the transposed view of an nn.Linear weight [N,K].
//   row_norm_kernel      attn[r] = |x[r,:]|_2, wave per row
//   topk_kernel          per image: bitonic sort of (attn, index) pairs in LDS (N <= 4096), descending, ties by lower
//                        index; gathers the first k rows of the features
//   l2_normalize_*       F.normalize along any dimension of a contiguous tensor (Whitener(l2norm=dim))
//   ln_act_f32_kernel    fp32 LayerNorm + erf GELU: the hidden layers of a multi-layer projector
//   spoc_kernel          per image: out = normalize(sum_n attn[n] feat[n,:])  (fp32 sums in token order, F.normalize eps 1e-12)
// Small problems (768 tokens x 1024 features per image): latency matters more than rate; fp64 MFMA peak is 78.6 TFLOP/s.
#include "common.hpp"
#include "kernels.hpp"

namespace m3r {

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f64x4 mfma4(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <class TC> struct Acc4;
template <> struct Acc4<double> { typedef f64x4 t; };
template <> struct Acc4<float> { typedef f32x4 t; };

struct GemmxArgs {
    const float* A;       // [M, K] fp32
    const void* sub;      // [K] TC or null
    const void* B;        // TC: [K, N] row-major, or [N, K] when b_transposed
    const void* bias;     // [N] TC or null
    const float* resid;   // [M, N] fp32 or null
    float* out;           // [M, N] fp32
    int M, N, K, b_transposed;
};

template <class TC>
__global__ void __launch_bounds__(256) gemmx_kernel(const GemmxArgs p) {
    typedef typename Acc4<TC>::t acc_t;
    constexpr int BM = 64, BN = 64, BK = 16;
    __shared__ TC As[BM][BK + 1];
    __shared__ TC Bs[BK][BN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const TC* __restrict__ B = reinterpret_cast<const TC*>(p.B);
    const TC* __restrict__ sub = reinterpret_cast<const TC*>(p.sub);
    acc_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
    const int fr = lane & 15, fk = lane >> 4;
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = e * 256 + tid;
            {   // A tile: 64 x 16, k fastest
                const int r = idx >> 4, k = idx & 15;
                const int gm = m0 + r, gk = k0 + k;
                TC v = 0;
                if (gm < p.M && gk < p.K) {
                    v = (TC)p.A[(size_t)gm * p.K + gk];
                    if (sub) v -= sub[gk];
                }
                As[r][k] = v;
            }
            {   // B tile: 16 x 64
                int k, n;
                if (p.b_transposed) { n = idx >> 4; k = idx & 15; }      // weight [N, K]: k fastest in memory
                else { k = idx >> 6; n = idx & 63; }                       // [K, N]: n fastest
                const int gk = k0 + k, gn = n0 + n;
                TC v = 0;
                if (gk < p.K && gn < p.N) v = p.b_transposed ? B[(size_t)gn * p.K + gk] : B[(size_t)gk * p.N + gn];
                Bs[k][n] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            TC a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[wm * 32 + i * 16 + fr][ks * 4 + fk];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[ks * 4 + fk][wn * 32 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma4(a[i], b[j], acc[i][j]);
        }
    }
    const TC* __restrict__ bias = reinterpret_cast<const TC*>(p.bias);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // D layout: f32 16x16x4 -> row 4*(lane/16) + r ; f64 16x16x4 -> row 4*r + lane/16 (register pairs)
                const int m = m0 + wm * 32 + i * 16 + (sizeof(TC) == 8 ? r * 4 + fk : fk * 4 + r);
                if (m < p.M && n < p.N) {
                    TC v = acc[i][j][r];
                    if (bias) v += bias[n];
                    float o = (float)v;
                    if (p.resid) o += p.resid[(size_t)m * p.N + n];
                    p.out[(size_t)m * p.N + n] = o;
                }
            }
        }
}

__global__ void __launch_bounds__(256) row_norm_kernel(const float* __restrict__ x, int M, int C, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = x[(size_t)row * C + c];
        s += v * v;
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = sqrtf(s);
}

// one block per image; N <= 4096 keys.  Sort key: attention descending, then index ascending (deterministic ties).
__global__ void __launch_bounds__(1024) topk_kernel(const float* __restrict__ feat, const float* __restrict__ attn, int N, int C, int k,
                                                    float* __restrict__ out_feat, float* __restrict__ out_attn, long long* __restrict__ out_idx) {
    __shared__ float key[4096];
    __shared__ int idx[4096];
    const int b = blockIdx.x;
    int P = 1;
    while (P < N) P <<= 1;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        key[i] = i < N ? attn[(size_t)b * N + i] : -INFINITY;
        idx[i] = i < N ? i : 0x7fffffff;
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < P / 2; i += blockDim.x) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);              // first half of each bitonic block sorted descending
                const float ka = key[lo], kb = key[hi];
                const int ia = idx[lo], ib = idx[hi];
                const bool a_first = (ka > kb) || (ka == kb && ia < ib);   // a ranks before b
                if (a_first != desc) {
                    key[lo] = kb; key[hi] = ka;
                    idx[lo] = ib; idx[hi] = ia;
                }
            }
            __syncthreads();
        }
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        out_attn[(size_t)b * k + j] = key[j];
        out_idx[(size_t)b * k + j] = idx[j];
    }
    for (int e = threadIdx.x; e < k * (C / 4); e += blockDim.x) {
        const int j = e / (C / 4), c4 = e - j * (C / 4);
        reinterpret_cast<f32x4*>(out_feat + ((size_t)b * k + j) * C)[c4] =
            reinterpret_cast<const f32x4*>(feat + ((size_t)b * N + idx[j]) * C)[c4];
    }
}

// one block per (image, 256-column slab)
__global__ void __launch_bounds__(256) spoc_sum_kernel(const float* __restrict__ feat, const float* __restrict__ attn, int N, int C,
                                                       float* __restrict__ out) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += feat[((size_t)b * N + n) * C + c] * attn[(size_t)b * N + n];
    out[(size_t)b * C + c] = s;
}
__global__ void __launch_bounds__(256) l2_normalize_rows_kernel(float* __restrict__ x, int M, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = x[(size_t)row * C + c];
        s += v * v;
    }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);      // F.normalize eps
    for (int c = lane; c < C; c += 64) x[(size_t)row * C + c] *= inv;
}

// F.normalize(x, dim) of a contiguous tensor seen as [outer, L, inner] (Whitener(l2norm=dim), retrieval/model.py:77-78): one thread per
// (outer, inner) column, fp32 sum of squares in index order, x / max(|x|_2, 1e-12).  inner == 1 takes the wave-per-row kernel above.
__global__ void __launch_bounds__(256) l2_normalize_strided_kernel(const float* x, long long outer, int L, long long inner,
                                                                   float* out) {   // out may alias x (no __restrict__)
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= outer * inner) return;
    const long long o = t / inner, i = t - o * inner;
    const float* src = x + o * L * inner + i;
    float* dst = out + o * L * inner + i;
    float s = 0.f;
    for (int l = 0; l < L; ++l) {
        const float v = src[(long long)l * inner];
        s += v * v;
    }
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    for (int l = 0; l < L; ++l) dst[(long long)l * inner] = src[(long long)l * inner] * inv;
}
__global__ void __launch_bounds__(256) l2_normalize_rows_copy_kernel(const float* x, long long M, int C, float* out) {   // out may alias x (no __restrict__)
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = x[row * C + c];
        s += v * v;
    }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int c = lane; c < C; c += 64) out[row * C + c] = x[row * C + c] * inv;
}

// nn.LayerNorm (two passes over the row held by one wave: mean, then centred variance) followed by nn.GELU (erf form, libm erff):
// the hidden layers of a multi-layer projector (retrieval/model.py:139-151: Linear - LayerNorm - GELU stacks), all fp32.
__global__ void __launch_bounds__(256) ln_act_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, int M, int C, int gelu,
                                                          float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] - mu;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    for (int c = lane; c < C; c += 64) {
        float v = (xr[c] - mu) * rstd;
        v = v * (gamma ? gamma[c] : 1.0f) + (beta ? beta[c] : 0.0f);
        if (gelu) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        out[(size_t)row * C + c] = v;
    }
}

int launch_l2_normalize(const float* x, long long outer, int L, long long inner, float* out, hipStream_t s, const char** err) {
    if (outer <= 0 || inner <= 0 || L <= 0) return 0;
    if (inner == 1) hipLaunchKernelGGL(l2_normalize_rows_copy_kernel, dim3((unsigned)((outer + 3) / 4)), dim3(256), 0, s, x, outer, L, out);
    else hipLaunchKernelGGL(l2_normalize_strided_kernel, dim3((unsigned)((outer * inner + 255) / 256)), dim3(256), 0, s, x, outer, L, inner, out);
    if (hipGetLastError() != hipSuccess) { *err = "l2_normalize: launch failed"; return 1; }
    return 0;
}

int launch_ln_act_f32(const float* x, const float* gamma, const float* beta, float eps, int M, int C, int gelu, float* out, hipStream_t s,
                      const char** err) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(ln_act_f32_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, gamma, beta, eps, M, C, gelu, out);
    if (hipGetLastError() != hipSuccess) { *err = "ln_act: launch failed"; return 1; }
    return 0;
}

int launch_gemmx(int is_double, const float* A, const void* sub, const void* B, int b_transposed, const void* bias, const float* resid,
                 float* out, int M, int N, int K, hipStream_t s, const char** err) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0) { *err = "gemmx: K must be positive"; return 1; }
    GemmxArgs a{A, sub, B, bias, resid, out, M, N, K, b_transposed};
    const dim3 grid((N + 63) / 64, (M + 63) / 64);
    if (is_double) hipLaunchKernelGGL(gemmx_kernel<double>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gemmx_kernel<float>, grid, dim3(256), 0, s, a);
    if (hipGetLastError() != hipSuccess) { *err = "gemmx: launch failed"; return 1; }
    return 0;
}

int launch_row_norm(const float* x, int M, int C, float* out, hipStream_t s, const char** err) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(row_norm_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, M, C, out);
    if (hipGetLastError() != hipSuccess) { *err = "row_norm: launch failed"; return 1; }
    return 0;
}

int launch_topk_gather(const float* feat, const float* attn, int Bn, int N, int C, int k, float* out_feat, float* out_attn,
                       long long* out_idx, hipStream_t s, const char** err) {
    if (Bn <= 0 || k <= 0) return 0;
    if (N > 4096 || k > N || (C % 4)) { *err = "topk: needs k <= N <= 4096 and C % 4 == 0"; return 1; }
    hipLaunchKernelGGL(topk_kernel, dim3(Bn), dim3(1024), 0, s, feat, attn, N, C, k, out_feat, out_attn, out_idx);
    if (hipGetLastError() != hipSuccess) { *err = "topk: launch failed"; return 1; }
    return 0;
}

int launch_weighted_spoc(const float* feat, const float* attn, int Bn, int N, int C, float* out, hipStream_t s, const char** err) {
    if (Bn <= 0) return 0;
    hipLaunchKernelGGL(spoc_sum_kernel, dim3((C + 255) / 256, Bn), dim3(256), 0, s, feat, attn, N, C, out);
    hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((Bn + 3) / 4), dim3(256), 0, s, out, Bn, C);
    if (hipGetLastError() != hipSuccess) { *err = "weighted_spoc: launch failed"; return 1; }
    return 0;
}

}  // namespace m3r
