// Probe: cycle stamps inside the K loop of gemm256p_kernel (NPH = 2, two barriers per phase) for one block: lane 0 of wave 0 (group 0) and of wave 4
// (group 1).  Chip-filling shapes, RANDOM operands (the chip's clock under MFMA load is data dependent, profiles/r03_pipe_rates.txt).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGEMM_TRACE scripts/probes/gemm256p_trace.hip -o scripts/probes/build/gemm256p_trace
#include "../../must3r_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace m3r;

static void fill_random(_Float16* d, size_t n, float scale) {
    std::vector<_Float16> h(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = (_Float16)((((s >> 8) & 0xffff) / 32768.0f - 1.0f) * scale);
    }
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

template <int WS, int BN>
static void run(int M, int N, int K, int trace_block) {
    _Float16 *A, *W, *out; float* bias;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2 * WS); hipMalloc(&out, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
    fill_random(A, (size_t)M * K, 1.0f); fill_random(W, (size_t)N * K * WS, 0.03f); hipMemset(bias, 0, N * 4);
    GemmArgs a;
    std::memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.wsplit = WS == 2 ? 2 : 0; a.batch = 1;
    a.trace_block = trace_block;
    const size_t lds = (size_t)2 * 4 * 128 * 64 * 2;
    auto kern = gemm256p_kernel<f16_t, EPI_STORE16, WS, BN, 2, 2>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = ((M + 255) / 256) * (N / BN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> t(2 * 64 * 8);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_gemm256_trace), t.size() * 8);
    const int nk = K / 64;
    printf("M %d N %d K %d  WS %d BN %d  grid %d block %d  %.1f us per launch (instrumented) = %.0f TF/s\n", M, N, K, WS, BN, grid, trace_block, ms * 100.f,
           2.0 * M * N * K / (ms * 1e-4) / 1e12 * (WS == 2 ? 1 : 1));
    for (int g = 0; g < 2; ++g) {
        printf("group %d   phase 0: issue(reads+dma) vmcnt lgkm barA mfma barB | phase 1: issue vmcnt lgkm barA mfma barB | K-tile total (s_memtime ticks)\n", g);
        const unsigned long long* b = &t[g * 64 * 8];
        const int last = nk < 30 ? nk - 2 : 29;
        double sum[13] = {0};
        int cnt = 0;
        for (int kt = 2; kt < last; ++kt) {
            const unsigned long long* r = &b[kt * 16];
            if (kt < 8 || kt % 4 == 0)
                printf("  %3d: %5llu %5llu %5llu %5llu %5llu %5llu | %5llu %5llu %5llu %5llu %5llu %5llu | %5llu\n", kt, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3],
                       r[5] - r[4], r[6] - r[5], r[7] - r[6], r[8] - r[7], r[9] - r[8], r[10] - r[9], r[11] - r[10], r[12] - r[11], b[(kt + 1) * 16] - r[0]);
            for (int i = 0; i < 12; ++i) sum[i] += (double)(r[i + 1] - r[i]);
            sum[12] += (double)(b[(kt + 1) * 16] - r[0]);
            ++cnt;
        }
        printf("  mean:");
        for (int i = 0; i < 13; ++i) printf(" %6.0f%s", sum[i] / cnt, (i == 5 || i == 11) ? " |" : "");
        printf("\n");
    }
    hipFree(A); hipFree(W); hipFree(out); hipFree(bias);
}

int main() {
    run<1, 256>(15360, 3072, 4096, 0);
    run<1, 256>(15360, 3072, 4096, 333);
    run<1, 256>(15360, 1024, 4096, 0);
    run<2, 128>(15360, 3072, 2048, 0);
    // fewer busy CUs: is the K-loop step the same when the chip draws less power / shares less L2 and fabric?
    run<1, 256>(2048, 3072, 4096, 0);    //  96 tiles
    run<1, 256>(256, 256, 4096, 0);      //   1 tile
    return 0;
}
