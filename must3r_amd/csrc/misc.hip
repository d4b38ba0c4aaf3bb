// HBM-bound row / elementwise kernels: LayerNorm (wave per row, fp32 statistics, 16-bit and/or fp32 output),
// patch im2col, casts, position grid, pointmap activation.  All vectorised to 16 B per lane where the
// layout allows; no LDS needed (wave shuffles only).
#include "common.hpp"
#include "kernels.hpp"
#include "options.hpp"
#include <cstring>
#include <cstdio>
#include <cstdlib>

namespace m3r {
// ------------------------------------------------------------------------------------------------------------------
// option registry (options.hpp)
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct OptRow { const char* name; long long def, lo, hi; };
const OptRow kOpts[OPT_COUNT] = {
    {"PERSIST", 0, 0, 1}, {"GEMM256", 1, 0, 2}, {"G256K", 1, 0, 2}, {"G256P", 1, 0, 1}, {"G256P_SPLIT", 1, 0, 2}, {"SPARSE_256", 1, 0, 1},
    {"SPARSE_LO", 1, 0, 1}, {"BK128", 1, 0, 1}, {"LN_ROWS", 1, 0, 1}, {"LNFOLD", 1, 0, 1}, {"ENC_CHUNK_ROWS", 32768, 256, 1 << 22}, {"ATTN_LZ", 1, 0, 2},
    {"LNFOLD256", 0, 0, 1}, {"G256_GM", 4, 1, 64},
};
long long g_opt_val[OPT_COUNT];
bool g_opt_init[OPT_COUNT];
}  // namespace

int opt(Opt o) {
    if (!g_opt_init[o]) {
        const OptRow& r = kOpts[o];
        long long v = r.def;
        char env[48];
        snprintf(env, sizeof(env), "M3R_%s", r.name);
        if (const char* e = getenv(env)) {
            char* end = nullptr;
            const long long x = strtoll(e, &end, 10);
            if (end == e || *end != 0 || x < r.lo || x > r.hi)
                fprintf(stderr, "libmust3r_hip: %s=%s ignored (an integer in [%lld, %lld] is expected); using %lld\n", env, e, r.lo, r.hi, r.def);
            else v = x;
        }
        g_opt_val[o] = v;
        g_opt_init[o] = true;
    }
    return (int)g_opt_val[o];
}

int opt_set(const char* name, long long value, const char** err) {
    static thread_local char msg[128];
    for (int o = 0; o < OPT_COUNT; ++o) {
        if (name == nullptr || strcmp(name, kOpts[o].name) != 0) continue;
        if (value < kOpts[o].lo || value > kOpts[o].hi) {
            snprintf(msg, sizeof(msg), "set_option: %s = %lld is outside [%lld, %lld]", name, value, kOpts[o].lo, kOpts[o].hi);
            *err = msg;
            return 1;
        }
        g_opt_val[o] = value;
        g_opt_init[o] = true;
        return 0;
    }
    snprintf(msg, sizeof(msg), "set_option: unknown option '%s'", name ? name : "(null)");
    *err = msg;
    return 1;
}


// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, C <= 1024, C % 4 == 0.  Two-pass statistics in registers.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) ln_kernel(const LnArgs p) {
    typedef typename Vec<T>::v4 v4;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const float* wp = p.w;
    const float* bp = p.b;
    const float* addp = p.add ? p.add + (size_t)row * p.C : nullptr;
    if (p.rows_per_group > 0) {
        const int g = row / p.rows_per_group;
        wp += (size_t)g * p.C;
        bp += (size_t)g * p.C;
        addp = (p.add && g < p.add_groups) ? p.add + (size_t)(row - g * p.rows_per_group) * p.C : nullptr;
    }
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < p.C) {
            if (p.x) v[i] = *reinterpret_cast<const f32x4*>(p.x + (size_t)row * p.C + c);
            else v[i] = __builtin_convertvector(*reinterpret_cast<const v4*>(reinterpret_cast<const T*>(p.x16) + (size_t)row * p.C + c), f32x4);
            if (addp) v[i] += *reinterpret_cast<const f32x4*>(addp + c);
            if (p.copy32) *reinterpret_cast<f32x4*>(p.copy32 + (size_t)row * p.C + c) = v[i];
            if (p.raw16) *reinterpret_cast<v4*>(reinterpret_cast<T*>(p.raw16) + (size_t)row * p.C + c) = cvt4<T>(v[i]);
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mean = wave_sum_dpp(s) / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < p.C) {
            v[i] -= mean;
            q += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
        }
    }
    const float rstd = rsqrtf(wave_sum_dpp(q) / (float)p.C + p.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < p.C) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wp + c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(bp + c);
            const f32x4 y = v[i] * rstd * w + b;
            if (p.out32) *reinterpret_cast<f32x4*>(p.out32 + (size_t)row * p.C + c) = y;
            if (p.out16) {
                const size_t ld = p.ld16 ? (size_t)p.ld16 : (size_t)p.C;
                const v4 h = cvt4<T>(y);
                *reinterpret_cast<v4*>(reinterpret_cast<T*>(p.out16) + row * ld + c) = h;
                if (p.out16_dup) *reinterpret_cast<v4*>(reinterpret_cast<T*>(p.out16_dup) + row * ld + c) = h;
                if (p.out16_lo) {
                    const f32x4 hf = __builtin_convertvector(h, f32x4);
                    *reinterpret_cast<v4*>(reinterpret_cast<T*>(p.out16_lo) + row * ld + c) = cvt4<T>(y - hf);
                }
            }
        }
    }
}

// r04: the same arithmetic per row (same bits), several rows per wave.  A wave of ln_kernel loads its row, reduces twice, loads the affine
// parameters, stores and exits: every row pays the full memory latency once, in series with two cross-lane reductions, and the launch is 7680-76800
// four-row blocks long.  Here a wave walks rows (stride = waves of the grid), requests row r + stride BEFORE it reduces row r, and keeps gamma / beta
// in registers (ungrouped launches).  NV = float4 chunks per lane (C <= 256 NV).
template <class T, int NV>
__global__ void __launch_bounds__(256) ln_rows_kernel(const LnArgs p) {
    typedef typename Vec<T>::v4 v4;
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * 4;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const bool grouped = p.rows_per_group > 0;
    bool act[NV];
    f32x4 w[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        act[i] = (i * 64 + lane) * 4 < p.C;
        w[i] = b[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (act[i] && !grouped) {
            w[i] = *reinterpret_cast<const f32x4*>(p.w + (i * 64 + lane) * 4);
            b[i] = *reinterpret_cast<const f32x4*>(p.b + (i * 64 + lane) * 4);
        }
    }
    auto load = [&](int r, f32x4 (&v)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (act[i]) {
                if (p.x) v[i] = *reinterpret_cast<const f32x4*>(p.x + (size_t)r * p.C + c);
                else v[i] = __builtin_convertvector(*reinterpret_cast<const v4*>(reinterpret_cast<const T*>(p.x16) + (size_t)r * p.C + c), f32x4);
            }
        }
    };
    f32x4 v[NV], nx[NV];
    load(row, v);
    for (; row < p.M; row += nwaves) {
        const int nr = row + nwaves;
        if (nr < p.M) load(nr, nx);
        const float* addp = p.add ? p.add + (size_t)row * p.C : nullptr;
        if (grouped) {
            const int g = row / p.rows_per_group;
            addp = (p.add && g < p.add_groups) ? p.add + (size_t)(row - g * p.rows_per_group) * p.C : nullptr;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (act[i]) {
                    w[i] = *reinterpret_cast<const f32x4*>(p.w + (size_t)g * p.C + (i * 64 + lane) * 4);
                    b[i] = *reinterpret_cast<const f32x4*>(p.b + (size_t)g * p.C + (i * 64 + lane) * 4);
                }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (act[i]) {
                if (addp) v[i] += *reinterpret_cast<const f32x4*>(addp + c);
                if (p.copy32) *reinterpret_cast<f32x4*>(p.copy32 + (size_t)row * p.C + c) = v[i];
                if (p.raw16) *reinterpret_cast<v4*>(reinterpret_cast<T*>(p.raw16) + (size_t)row * p.C + c) = cvt4<T>(v[i]);
                s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
        }
        const float mean = wave_sum_dpp(s) / (float)p.C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (act[i]) {
                v[i] -= mean;
                q += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
            }
        const float rstd = rsqrtf(wave_sum_dpp(q) / (float)p.C + p.eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (act[i]) {
                const f32x4 y = v[i] * rstd * w[i] + b[i];
                if (p.out32) *reinterpret_cast<f32x4*>(p.out32 + (size_t)row * p.C + c) = y;
                if (p.out16) {
                    const size_t ld = p.ld16 ? (size_t)p.ld16 : (size_t)p.C;
                    const v4 h = cvt4<T>(y);
                    *reinterpret_cast<v4*>(reinterpret_cast<T*>(p.out16) + row * ld + c) = h;
                    if (p.out16_dup) *reinterpret_cast<v4*>(reinterpret_cast<T*>(p.out16_dup) + row * ld + c) = h;
                    if (p.out16_lo) {
                        const f32x4 hf = __builtin_convertvector(h, f32x4);
                        *reinterpret_cast<v4*>(reinterpret_cast<T*>(p.out16_lo) + row * ld + c) = cvt4<T>(y - hf);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = nx[i];
    }
}

// M3R_LN_ROWS (A/B instrument): 0 = one row per wave (ln_kernel, r01-r03), 1 = row-walking waves with the next row in flight (default)
static int ln_rows_mode() { return opt(OPT_LN_ROWS); }
template <class T>
static void launch_ln_t(const LnArgs& a, hipStream_t s) {
    // Measured (profiles/r04_ln_rows_ab.txt, same bits): the walk wins where the rows stream from HBM -- the render batch, 307200 x 768:
    // 324 -> 276 us, 4.4 -> 5.1 TB/s -- and loses where the producer GEMM left them in the 256 MB MALL -- an encoder chunk, 30720 x 1024:
    // 32.4 -> 39.2 us (5.8 -> 4.8 TB/s: 194 VGPRs = 2 walkers per SIMD cannot keep as many lines in flight as 8 one-row waves can).
    // Hence: only launches whose rows cannot be MALL-resident (> 64 k rows = 300+ MB).
    const int grid_one = (a.M + 3) / 4;
    if (ln_rows_mode() == 0 || a.M < 65536) {
        hipLaunchKernelGGL(ln_kernel<T>, dim3(grid_one), dim3(256), 0, s, a);
        return;
    }
    const int grid = 256 * 8;   // 8 resident blocks per CU (32 waves): every wave slot of the chip holds one walker
    if (a.C <= 768) hipLaunchKernelGGL((ln_rows_kernel<T, 3>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((ln_rows_kernel<T, 4>), dim3(grid), dim3(256), 0, s, a);
}

int launch_layernorm(DType dt, const LnArgs& a, hipStream_t s, const char** err) {
    if (a.M <= 0) return 0;
    if (a.C > 1024 || a.C % 4) { *err = "layernorm: C must be <= 1024 and a multiple of 4"; return 1; }
    if (dt == DT_BF16) launch_ln_t<bf16_t>(a, s);
    else launch_ln_t<f16_t>(a, s);
    if (hipGetLastError() != hipSuccess) { *err = "layernorm: launch failed"; return 1; }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// im2col for the 16x16/16 patch embedding: out[(v*N + t)*768 + c*256 + i*16 + j] = img[v][c][16gy+i][16gx+j]
// one thread = 8 consecutive j (two float4 loads, one 16-byte store)
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void im2col_kernel(const float* __restrict__ img, T* __restrict__ out, int V, int H, int W) {
    typedef typename Vec<T>::v8 v8;
    const int gh = H / 16, gw = W / 16;
    const size_t total = (size_t)V * gh * gw * 96;  // 768/8 chunks per token
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % 96);
        const size_t tok = idx / 96;
        const int t = (int)(tok % (gh * gw));
        const int v = (int)(tok / (gh * gw));
        const int gy = t / gw, gx = t - gy * gw;
        const int c = ch / 32, rem = ch - c * 32;
        const int i = rem >> 1, j0 = (rem & 1) * 8;
        const float* src = img + (((size_t)v * 3 + c) * H + gy * 16 + i) * W + gx * 16 + j0;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src);
        const f32x4 b = *reinterpret_cast<const f32x4*>(src + 4);
        const f32x8 f = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<v8*>(out + tok * 768 + ch * 8) = cvt8<T>(f);
    }
}

int launch_im2col(DType dt, const float* img, void* out16, int V, int H, int W, hipStream_t s, const char** err) {
    if (H % 16 || W % 16) { *err = "im2col: H and W must be multiples of 16"; return 1; }
    const size_t total = (size_t)V * (H / 16) * (W / 16) * 96;
    if (!total) return 0;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dt == DT_BF16) hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, img, (bf16_t*)out16, V, H, W);
    else hipLaunchKernelGGL(im2col_kernel<f16_t>, dim3(grid), dim3(256), 0, s, img, (f16_t*)out16, V, H, W);
    if (hipGetLastError() != hipSuccess) { *err = "im2col: launch failed"; return 1; }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fp32 -> 16-bit (hi) and optional lo = T(x - float(hi))
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ hi, T* __restrict__ lo, size_t n4) {
    typedef typename Vec<T>::v4 v4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 x = reinterpret_cast<const f32x4*>(in)[i];
        const v4 h = cvt4<T>(x);
        reinterpret_cast<v4*>(hi)[i] = h;
        if (lo) reinterpret_cast<v4*>(lo)[i] = cvt4<T>(x - __builtin_convertvector(h, f32x4));
    }
}

int launch_split16(DType dt, const float* in, void* hi, void* lo, size_t n, hipStream_t s, const char** err) {
    if (n % 4) { *err = "cast: element count must be a multiple of 4"; return 1; }
    if (!n) return 0;
    const size_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    if (dt == DT_BF16) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, in, (bf16_t*)hi, (bf16_t*)lo, n4);
    else hipLaunchKernelGGL(cast_kernel<f16_t>, dim3(grid), dim3(256), 0, s, in, (f16_t*)hi, (f16_t*)lo, n4);
    if (hipGetLastError() != hipSuccess) { *err = "cast: launch failed"; return 1; }
    return 0;
}

int launch_cast(DType dt, const float* in, void* out16, void* out16_lo, size_t n, hipStream_t s, const char** err) {
    return launch_split16(dt, in, out16, out16_lo, n, s, err);
}

// ------------------------------------------------------------------------------------------------
// 2:4-sparse copy of the fp16 low part of a weight matrix (r05; consumed by gemm256p_kernel<.., WS = 3>, GemmArgs::Wlo_sp / Widx_sp).
//   lo = fp16(w - float(fp16(w))) exactly as cast_kernel computes it; in every group of 4 consecutive k (k = 64 t + 16 g + 4 q + e) of a row the two entries
//   of largest |lo| are kept (ties: the lower k), in k order.  One thread = one (K-tile t, row n, lane group g): 16 low parts -> 8 kept values
//   (vals[t][n][8 g .. 8 g + 7]) and 16 bits of positions (group q: bits 4q+1..4q = position of the first kept entry, 4q+3..4q+2 = of the second --
//   the index operand of v_smfmac_f32_16x16x64_f16, profiles/r05_smfmac_probe.txt) at halfword (n % 32) / 16 of dword idx[t][n / 32][n % 16 + 16 g].
// ------------------------------------------------------------------------------------------------
__global__ void sparse24_pack_kernel(const float* __restrict__ w, int rows, int K, f16_t* __restrict__ vals, unsigned short* __restrict__ idx) {
    const size_t total = (size_t)(K / 64) * rows * 4;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(u & 3);
        const size_t tn = u >> 2;
        const int n = (int)(tn % rows), t = (int)(tn / rows);
        const float* src = w + (size_t)n * K + t * 64 + g * 16;
        unsigned bits = 0;
        f16_t out[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float lo[4], mag[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = src[q * 4 + e];
                const f16_t hi = (f16_t)x;
                lo[e] = (float)(f16_t)(x - (float)hi);
                mag[e] = fabsf(lo[e]);
            }
            int p0 = 0;
#pragma unroll
            for (int e = 1; e < 4; ++e) if (mag[e] > mag[p0]) p0 = e;
            int p1 = p0 == 0 ? 1 : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e != p0 && mag[e] > mag[p1]) p1 = e;
            const int a = p0 < p1 ? p0 : p1, b = p0 < p1 ? p1 : p0;
            out[q * 2] = (f16_t)lo[a];
            out[q * 2 + 1] = (f16_t)lo[b];
            bits |= (unsigned)(a | (b << 2)) << (4 * q);
        }
        f16_t* dv = vals + ((size_t)t * rows + n) * 32 + g * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dv[e] = out[e];
        idx[(((size_t)t * (rows / 32) + n / 32) * 64 + (n % 16) + 16 * g) * 2 + ((n % 32) / 16)] = (unsigned short)bits;
    }
}

int launch_sparse24_pack(const float* w, int rows, int K, void* vals, void* idx, hipStream_t s, const char** err) {
    if (rows <= 0 || K <= 0 || rows % 32 || K % 64) { *err = "sparse24_pack: rows % 32 and K % 64 must be 0"; return 1; }
    const size_t total = (size_t)(K / 64) * rows * 4;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(sparse24_pack_kernel, dim3(grid), dim3(256), 0, s, w, rows, K, (f16_t*)vals, (unsigned short*)idx);
    if (hipGetLastError() != hipSuccess) { *err = "sparse24_pack: launch failed"; return 1; }
    return 0;
}

#ifdef M3R_ATTN_FP8   // parked with the e4m3 attention path (kernels.hpp)
// ------------------------------------------------------------------------------------------------
// 16-bit -> OCP e4m3 (fp8 attention operands, include/must3r_hip.h MUST3R_ATTN_FP8): out8[r][c] = e4m3(clamp(in[r][c], +-448)).
// v_cvt_pk_fp8_f32 does NOT saturate (1000 -> NaN, profiles/r02_fp8_probe.txt), hence the clamp.  One thread = 8 columns
// (16 B read, 8 B written).  Rows may be grouped: row r of group g = r / rows_per_group goes to out_table[g] (the per-layer
// memory buffers of the grouped post-feedback K|V projection).
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void quant8_kernel(const T* __restrict__ in, int ld_in, unsigned char* __restrict__ out, int ld_out, void* const* out_table,
                              int rows_per_group, size_t rows, int cols, int tail_cols) {
    typedef typename Vec<T>::v8 v8;
    const int cpr = (cols + tail_cols) / 8;
    const size_t total = rows * (size_t)cpr;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / cpr;
        const int c = (int)(idx - r * cpr) * 8;
        const v8 raw = *reinterpret_cast<const v8*>(in + r * ld_in + c);
        unsigned char* dst = out;
        size_t rr = r;
        if (out_table) {
            const size_t g = r / rows_per_group;
            dst = reinterpret_cast<unsigned char*>(out_table[g]);
            rr = r - g * rows_per_group;
        }
        if (c >= cols) {   // 16-bit tail, copied as is
            *reinterpret_cast<v8*>(dst + rr * ld_out + cols + (size_t)(c - cols) * 2) = raw;
            continue;
        }
        const f32x8 x = __builtin_convertvector(raw, f32x8);
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = __builtin_amdgcn_fmed3f(x[e], -448.0f, 448.0f);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[4], y[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[6], y[7], hi, true);
        *reinterpret_cast<u32x2*>(dst + rr * ld_out + c) = u32x2{(unsigned)lo, (unsigned)hi};
    }
}

int launch_quant8(DType dt, const void* in16, int ld_in, void* out8, int ld_out, void* const* out_table, int rows_per_group,
                  size_t rows, int cols, int tail_cols, hipStream_t s, const char** err) {
    if (!rows || !(cols + tail_cols)) return 0;
    if (cols % 16 || tail_cols % 8 || ld_in % 8 || ld_out % 16) { *err = "quant8: cols % 16, tail_cols % 8, ld_in % 8, ld_out % 16"; return 1; }
    const size_t total = rows * (size_t)((cols + tail_cols) / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dt == DT_BF16) hipLaunchKernelGGL(quant8_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)in16, ld_in, (unsigned char*)out8, ld_out, out_table, rows_per_group, rows, cols, tail_cols);
    else hipLaunchKernelGGL(quant8_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (const f16_t*)in16, ld_in, (unsigned char*)out8, ld_out, out_table, rows_per_group, rows, cols, tail_cols);
    if (hipGetLastError() != hipSuccess) { *err = "quant8: launch failed"; return 1; }
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------------
// positions: pos[v][gy*gw+gx] = (gy, gx)   (croco PositionGetter)
// ------------------------------------------------------------------------------------------------
__global__ void fill_pos_kernel(int64_t* pos, int V, int gh, int gw) {
    const size_t total = (size_t)V * gh * gw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % (gh * gw));
        pos[i * 2 + 0] = t / gw;
        pos[i * 2 + 1] = t % gw;
    }
}

int launch_fill_pos(int64_t* pos, int V, int gh, int gw, hipStream_t s, const char** err) {
    const size_t total = (size_t)V * gh * gw;
    if (!total) return 0;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(fill_pos_kernel, dim3(grid), dim3(256), 0, s, pos, V, gh, gw);
    if (hipGetLastError() != hipSuccess) { *err = "fill_pos: launch failed"; return 1; }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// pointmap activation (engine/inference.py:19-27, tools/geometry.py:14-18):
//   pts3d = v/max(|v|,1e-8) * expm1(|v|) on ch 0:3, same on ch 3:6, conf = 1 + exp(ch 6)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void norm_exp3(const float* v, float* o) {
    const float d = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float sc = expm1f(d) / fmaxf(d, 1e-8f);
    o[0] = v[0] * sc;
    o[1] = v[1] * sc;
    o[2] = v[2] * sc;
}

template <bool LINEAR>
__global__ void postprocess_kernel(const float* __restrict__ pm, float* __restrict__ p3, float* __restrict__ pl,
                                   float* __restrict__ cf, size_t npix) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        float v[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k] = pm[i * 7 + k];
        float a[3], b[3];
        if constexpr (LINEAR) {   // ActivationType.LINEAR (head.py:13-21): the coordinates pass through
#pragma unroll
            for (int k = 0; k < 3; ++k) { a[k] = v[k]; b[k] = v[3 + k]; }
        } else {
            norm_exp3(v, a);
            norm_exp3(v + 3, b);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p3[i * 3 + k] = a[k];
            pl[i * 3 + k] = b[k];
        }
        cf[i] = 1.0f + expf(v[6]);
    }
}

int launch_postprocess(const float* pm, int linear, float* pts3d, float* pts3d_local, float* conf, size_t npix, hipStream_t s,
                       const char** err) {
    if (!npix) return 0;
    const int grid = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
    if (linear) hipLaunchKernelGGL(postprocess_kernel<true>, dim3(grid), dim3(256), 0, s, pm, pts3d, pts3d_local, conf, npix);
    else hipLaunchKernelGGL(postprocess_kernel<false>, dim3(grid), dim3(256), 0, s, pm, pts3d, pts3d_local, conf, npix);
    if (hipGetLastError() != hipSuccess) { *err = "postprocess: launch failed"; return 1; }
    return 0;
}

}  // namespace m3r
