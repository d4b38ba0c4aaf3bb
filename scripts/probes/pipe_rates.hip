// Probe: what does one CU sustain per "K-tile" of a small-M GEMM?  A block of NW waves loops; per iteration every wave issues
// P LDS-DMA pieces (1 KB each, from an L2-resident region private to the block), R ds_read_b128 and M MFMAs (16x16x32 f16), then
// waits for its DMA (counted: one iteration stays in flight) and passes a barrier -- the structure of gemm96 / gemm48 / the 64x64
// ring without any address arithmetic in the loop.  Prints shader cycles per iteration (s_memtime of wave 0) and the wall time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/pipe_rates.hip -o scripts/probes/build/pipe_rates && scripts/probes/build/pipe_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int P, int R, int M, int BAR>
__global__ void __launch_bounds__(1024) probe(const char* __restrict__ src, int iters, unsigned long long* cyc, float* sink, int stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    // per-block 256 KB region (L2 resident); piece p of wave w in iteration it: contiguous 1 KB chunks, cycling over 4 slots
    // stride 0: a piece is 1 KB contiguous; else 8 rows x 128 B at `stride` bytes (the GEMM operand tiles)
    const char* base = src + (size_t)blockIdx.x * (256 << 10) + (stride ? (lane >> 3) * stride + (lane & 7) * 16 : lane * 16);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    f16x8 rd[4];
    for (int i = 0; i < 4; ++i) rd[i] = a;
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + lane * 16;
    unsigned long long t0 = 0;
    for (int it = 0; it < iters; ++it) {
        if (it == 8) t0 = __builtin_readcyclecounter();
        const int slot = it & 3;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int piece = (slot * P + p) * nw + wave;   // < 4 * P * nw <= 144 KB
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (stride ? (size_t)((it * P + p) & 7) * 128 + (size_t)(wave & 7) * 8 * stride : (size_t)((it * P + p) & 63) * 4096 + wave * 128)),
                                             (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f16x8 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lbase + (unsigned)(wave * 4096)), "n"((r & 3) * 1024));
            rd[r & 3] = v;
        }
        if (R) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rd[m & 3], b, acc[m & 3], 0, 0, 0);
        if (P) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + (float)rd[i][0];
    if (s == 12345.678f) sink[0] = s;
}

// r03: the same loop with the DMA pieces shaped like a GEMM operand tile of row pitch `stride`: a 1 KB piece is (1024 / ROWB) rows x ROWB
// bytes (ROWB = 64: K-tile of 32 halves -- half a cache line per row; ROWB = 128: K-tile of 64 halves), advancing ROWB bytes along K per
// iteration; the 32 blocks of an XCD share one L2-resident 1 MB region.
template <int P, int R, int M, int BAR, int ROWB>
__global__ void __launch_bounds__(1024) probe_rows(const char* __restrict__ src, int iters, unsigned long long* cyc, float* sink, int stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    constexpr int LPR = ROWB / 16, RPP = 1024 / ROWB;
    const char* base = src + (size_t)(blockIdx.x & 7) * (1 << 20) + (size_t)(lane / LPR) * stride + (lane % LPR) * 16;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    f16x8 rd[4];
    for (int i = 0; i < 4; ++i) rd[i] = a;
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + lane * 16;
    unsigned long long t0 = 0;
    for (int it = 0; it < iters; ++it) {
        if (it == 8) t0 = __builtin_readcyclecounter();
        const int slot = it & 3;
        const int koff = (it * ROWB) & (stride - 1);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int piece = (slot * P + p) * nw + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + koff + (size_t)((wave * P + p) * RPP) * stride),
                                             (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f16x8 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lbase + (unsigned)(wave * 4096)), "n"((r & 3) * 1024));
            rd[r & 3] = v;
        }
        if (R) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rd[m & 3], b, acc[m & 3], 0, 0, 0);
        if (P) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + (float)rd[i][0];
    if (s == 12345.678f) sink[0] = s;
}

template <int P, int R, int M, int BAR, int ROWB>
void run_rows(const char* buf, int nw, unsigned long long* cyc, float* sink, int stride) {
    const int iters = 2008;
    const size_t ldsb = 150 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_rows<P, R, M, BAR, ROWB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe_rows<P, R, M, BAR, ROWB><<<256, nw * 64, ldsb>>>(buf, 64, cyc, sink, stride);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe_rows<P, R, M, BAR, ROWB><<<256, nw * 64, ldsb>>>(buf, iters, cyc, sink, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)c / (iters - 8);
    printf("rows of %3d B, pitch %5d, waves %2d  DMA %2d KB  reads %3d KB  MFMA/wave %2d  barrier %d : %7.0f cycles/iter  %6.3f us/iter  (DMA %5.1f B/clk, MFMA pipe %4.0f%%)\n",
           ROWB, stride, nw, P * nw, R * nw, M, BAR, per, ms * 1e3 / iters, P * nw * 1024.0 / per, 100.0 * M * nw / 4.0 * 16 / per);
}

// r03: MFMA-only loop (2 waves per SIMD, 32 MFMAs per wave and iteration, 16 independent accumulators) on operands that are all ones,
// random fp16 values held in a few registers, or random values in MANY registers (8 A x 4 B fragments cycling like a GEMM wave tile):
// is the matrix pipe's sustained rate data dependent (power)?
template <int MODE>
__global__ void __launch_bounds__(512) probe_mfma(const _Float16* __restrict__ rnd, int iters, unsigned long long* cyc, float* sink) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 8; ++e) a[i][e] = MODE == 0 ? (_Float16)1 : rnd[((MODE == 1 ? 0 : i) * 64 + lane) * 8 + e];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) b[i][e] = MODE == 0 ? (_Float16)1 : rnd[4096 + ((MODE == 1 ? 0 : i) * 64 + lane) * 8 + e];
    unsigned long long t0 = 0;
    for (int it = 0; it < iters; ++it) {
        if (it == 8) t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int m = 0; m < 32; ++m) acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m & 7], b[(m >> 3) & 3], acc[m & 15], 0, 0, 0);
        // keep the accumulators bounded (random data would overflow to inf and the multipliers would see constant operands)
        if ((it & 63) == 63)
            for (int i = 0; i < 16; ++i) acc[i] *= 1e-3f;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    if (s == 12345.678f) sink[0] = s;
}
template <int MODE>
void run_mfma(const _Float16* rnd, unsigned long long* cyc, float* sink) {
    const int iters = 20008;   // ~10 ms: long enough for the power management to settle
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe_mfma<MODE><<<256, 512>>>(rnd, 64, cyc, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe_mfma<MODE><<<256, 512>>>(rnd, iters, cyc, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double us = ms * 1e3 / iters;
    printf("MFMA only, operands %-32s: %6.0f cycles/iter  %6.3f us/iter  = %6.0f TF/s on 256 CUs\n",
           MODE == 0 ? "all ones" : MODE == 1 ? "random, one A / one B fragment" : "random, 8 A x 4 B fragments", (double)c / (iters - 8), us,
           256.0 * 8 * 32 * 16384 / us * 1e-6);
}

// r03: a K-tile whose DMA pieces are a MIX of 64-byte-row and 128-byte-row gathers (the split-weight 256 x 256 x 32 K-tile: activations 2 pieces of
// 16 rows x 64 B per wave; weights hi+lo either 4 more such pieces, or -- with hi and lo interleaved per 32 k in memory -- 4 pieces of 8 rows x 128 B)
template <int P64, int P128, int R, int M, int BAR>
__global__ void __launch_bounds__(1024) probe_mix(const char* __restrict__ src, int iters, unsigned long long* cyc, float* sink, int stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    constexpr int P = P64 + P128;
    const char* base64 = src + (size_t)(blockIdx.x & 7) * (1 << 20) + (size_t)(lane / 4) * stride + (lane % 4) * 16;
    const char* base128 = src + (size_t)(blockIdx.x & 7) * (1 << 20) + (size_t)(lane / 8) * stride + (lane % 8) * 16;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    f16x8 rd[4];
    for (int i = 0; i < 4; ++i) rd[i] = a;
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + lane * 16;
    for (int it = 0; it < iters; ++it) {
        const int slot = it & 1;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int piece = (slot * P + p) * nw + wave;
            const char* g = p < P64 ? base64 + ((it * 64) & (stride - 1)) + (size_t)((wave * P + p) * 16 % 448) * stride
                                    : base128 + ((it * 128) & (stride - 1)) + (size_t)((wave * P + p) * 8 % 448) * stride;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f16x8 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lbase + (unsigned)(wave * 4096)), "n"((r & 3) * 1024));
            rd[r & 3] = v;
        }
        if (R) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rd[m & 3], b, acc[m & 3], 0, 0, 0);
        if (P) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + (float)rd[i][0];
    if (s == 12345.678f) sink[0] = s;
}
template <int P64, int P128, int R, int M, int BAR>
void run_mix(const char* buf, unsigned long long* cyc, float* sink) {
    const int iters = 4000;
    const size_t ldsb = 150 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_mix<P64, P128, R, M, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe_mix<P64, P128, R, M, BAR><<<256, 512, ldsb>>>(buf, 64, cyc, sink, 2048);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe_mix<P64, P128, R, M, BAR><<<256, 512, ldsb>>>(buf, iters, cyc, sink, 2048);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("8 waves: %d pieces of 64 B rows + %d pieces of 128 B rows per wave (%2d KB), %2d reads, %2d MFMAs per wave, barrier %d : %6.3f us/iter\n", P64, P128,
           (P64 + P128) * 8, R, M, BAR, ms * 1e3 / iters);
}

template <int P, int R, int M, int BAR>
void run(const char* buf, int nw, unsigned long long* cyc, float* sink, int stride = 0) {
    const int iters = 2008;
    const size_t ldsb = 150 * 1024;   // one block per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<P, R, M, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<P, R, M, BAR><<<256, nw * 64, ldsb>>>(buf, 64, cyc, sink, stride);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<P, R, M, BAR><<<256, nw * 64, ldsb>>>(buf, iters, cyc, sink, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)c / (iters - 8);
    printf("stride %5d waves %2d  DMA %2d KB  reads %3d KB  MFMA/wave %2d  barrier %d : %7.0f cycles/iter  %6.3f us/iter  (DMA %5.1f B/clk, LDS rd %5.1f B/clk, MFMA pipe %4.0f%%)\n",
           stride, nw, P * nw, R * nw, M, BAR, per, ms * 1e3 / iters, P * nw * 1024.0 / per, R * nw * 1024.0 / per, 100.0 * M * nw / 4.0 * 16 / per);
}

int main() {
    char* buf; float* sink; unsigned long long* cyc;
    hipMalloc(&buf, 256ull * (256 << 10)); hipMalloc(&sink, 64); hipMalloc(&cyc, 64);
    hipMemset(buf, 0x3c, 256ull * (256 << 10));
    {
        std::vector<_Float16> h(8192);
        unsigned x = 12345;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 8) % 2001 - 1000) * 1e-3f); }
        _Float16* rnd; hipMalloc(&rnd, 8192 * 2); hipMemcpy(rnd, h.data(), 8192 * 2, hipMemcpyHostToDevice);
        run_mfma<0>(rnd, cyc, sink); run_mfma<1>(rnd, cyc, sink); run_mfma<2>(rnd, cyc, sink); run_mfma<0>(rnd, cyc, sink);
    }
    printf("-- split-weight K-tile (256 x 256 x 32, hi + lo): weight pieces as 64 B rows (today) or as 128 B rows (hi / lo interleaved per 32 k)\n");
    run_mix<6, 0, 0, 0, 1>(buf, cyc, sink); run_mix<2, 4, 0, 0, 1>(buf, cyc, sink);
    run_mix<6, 0, 16, 64, 1>(buf, cyc, sink); run_mix<2, 4, 16, 64, 1>(buf, cyc, sink);
    run_mix<6, 0, 16, 64, 0>(buf, cyc, sink); run_mix<2, 4, 16, 64, 0>(buf, cyc, sink);
    // r03: the chip-filling 256 x 256 x 32 K-tile (8 waves: 4 DMA pieces, 12 fragment reads, 32 (plain) / 64 (split: 6 pieces, 16 reads) MFMAs per wave)
    printf("-- gemm256 K-tile mix, 8 waves\n");
    run<0, 0, 32, 1>(buf, 8, cyc, sink, 2048);
    run<4, 0, 0, 1>(buf, 8, cyc, sink, 2048);
    run<0, 12, 0, 1>(buf, 8, cyc, sink, 2048);
    run<4, 12, 32, 1>(buf, 8, cyc, sink, 2048);
    run<4, 12, 32, 0>(buf, 8, cyc, sink, 2048);
    run<2, 8, 32, 1>(buf, 8, cyc, sink, 2048);     // weight fragments not through LDS: half the DMA, 8 reads
    run<2, 8, 32, 0>(buf, 8, cyc, sink, 2048);
    run<6, 16, 64, 1>(buf, 8, cyc, sink, 2048);
    run<6, 16, 64, 0>(buf, 8, cyc, sink, 2048);
    run<2, 8, 64, 1>(buf, 8, cyc, sink, 2048);
    printf("-- operand-tile shaped DMA pieces (8 waves): 64 B rows (K-tile 32) against 128 B rows (K-tile 64)\n");
    for (int pitch : {2048, 8192}) {
        run_rows<4, 0, 0, 1, 64>(buf, 8, cyc, sink, pitch);
        run_rows<4, 0, 0, 1, 128>(buf, 8, cyc, sink, pitch);
        run_rows<4, 12, 32, 1, 64>(buf, 8, cyc, sink, pitch);
        run_rows<4, 12, 32, 1, 128>(buf, 8, cyc, sink, pitch);
        run_rows<4, 12, 32, 0, 64>(buf, 8, cyc, sink, pitch);
        run_rows<4, 12, 32, 0, 128>(buf, 8, cyc, sink, pitch);
    }
    printf("-- 4 waves, 128 x 128 wave tiles (8 DMA pieces, 16 reads, 64 MFMAs per wave)\n");
    run<8, 16, 64, 1>(buf, 4, cyc, sink, 2048);
    run<8, 16, 64, 0>(buf, 4, cyc, sink, 2048);
    if (getenv("PIPE_R03_ONLY")) return 0;
    for (int stride : {0, 1536, 3072, 2048, 1664}) {
        printf("-- stride %d, 9 waves\n", stride);
        run<4, 0, 0, 1>(buf, 9, cyc, sink, stride);
        run<4, 12, 16, 1>(buf, 9, cyc, sink, stride);
        run<3, 10, 8, 1>(buf, 8, cyc, sink, stride);
    }
    return 0;
}
