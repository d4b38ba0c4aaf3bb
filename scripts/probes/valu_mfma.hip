// Probe (r03): do VALU instructions and MFMAs of DIFFERENT waves of one SIMD overlap, and what does v_exp_f32 cost?
// NW waves per SIMD loop over: M MFMAs (16x16x32 f16, 16 independent accumulators), E v_exp_f32 and V v_fma_f32 (32 independent chains),
// either as separate phases (MFMAs, then VALU -- the order of a flash-attention tile) or finely interleaved.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/valu_mfma.hip -o scripts/probes/build/valu_mfma && scripts/probes/build/valu_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int M, int E, int V, int MIX>
__global__ void __launch_bounds__(1024) probe(int iters, float* sink, float seed) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x + 3 * e)); }
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = seed + 0.001f * i;
    for (int it = 0; it < iters; ++it) {
        if (MIX == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 15], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < E; ++e) x[e & 31] = __builtin_amdgcn_exp2f(x[e & 31]);
#pragma unroll
            for (int v = 0; v < V; ++v) x[v & 31] = __builtin_fmaf(x[v & 31], 0.999f, 0.001f);
        } else {
            constexpr int N = M > E + V ? M : E + V;
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (n * M / N != (n + 1) * M / N || (n == 0 && M > 0 && M >= N)) acc[n & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[n & 15], 0, 0, 0);
                if (n < E) x[n & 31] = __builtin_amdgcn_exp2f(x[n & 31]);
                else if (n < E + V) x[n & 31] = __builtin_fmaf(x[n & 31], 0.999f, 0.001f);
            }
        }
        if (E > 0) {   // keep the exp2 arguments in range
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(x[i]));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    for (int i = 0; i < 32; ++i) s += x[i];
    if (s == 12345.678f) sink[0] = s;
}

template <int M, int E, int V, int MIX>
void run(int waves_per_simd, float* sink) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<M, E, V, MIX><<<256, waves_per_simd * 256>>>(64, sink, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<M, E, V, MIX><<<256, waves_per_simd * 256>>>(iters, sink, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("waves/SIMD %d  MFMA %2d  exp %2d  fma %2d  %s : %7.1f ns per iteration (all waves of a SIMD)\n", waves_per_simd, M, E, V,
           MIX ? "interleaved" : "phases     ", ms * 1e6 / iters);
}

int main() {
    float* sink; hipMalloc(&sink, 64);
    for (int w : {1, 2, 3}) {
        printf("-- %d wave(s) per SIMD\n", w);
        run<36, 0, 0, 0>(w, sink);
        run<0, 32, 0, 0>(w, sink);
        run<0, 0, 32, 0>(w, sink);
        run<0, 0, 64, 0>(w, sink);
        run<0, 32, 33, 0>(w, sink);
        run<36, 32, 0, 0>(w, sink);
        run<36, 0, 32, 0>(w, sink);
        run<36, 0, 64, 0>(w, sink);
        run<36, 32, 33, 0>(w, sink);
        run<36, 32, 33, 1>(w, sink);
        run<36, 0, 64, 1>(w, sink);
    }
    return 0;
}
