// Probe: cycle stamps inside the K loop of gemm256_kernel (8 waves, two groups staggered by one barrier) for block 0: lane 0 of wave 0
// (group 0) and of wave 4 (group 1).  Chip-filling shape (all 256 CUs busy, operands streamed through L2 / MALL like in the scene).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGEMM_TRACE scripts/probes/gemm256_trace.hip -o scripts/probes/build/gemm256_trace
#include "../../must3r_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace m3r;

template <int BN>
static void run(int M, int N, int K) {
    _Float16 *A, *W, *out; float* bias;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2 * 2); hipMalloc(&out, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
    hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(W, 0x2c, (size_t)N * K * 4); hipMemset(bias, 0, N * 4);
    GemmArgs a;
    std::memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.wsplit = 2; a.batch = 1;
    constexpr int NST = 2 * BN >= 384 ? 3 : 4;
    const size_t lds = (size_t)NST * (256 + 2 * BN) * 32 * 2;
    auto kern = gemm256_kernel<f16_t, EPI_STORE16, 2, BN, 1>;
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
    const int nk = K / 32;
    printf("M %d N %d K %d  BN %d  grid %d  %.1f us per launch (instrumented)\n", M, N, K, BN, grid, ms * 100.f);
    for (int g = 0; g < 2; ++g) {
        printf("group %d  step: wait_dma barrier1 reads+dma_issue wait_dma(g1) lgkmcnt barrier2 mfma_issue | step total (cycle counter ticks)\n", g);
        const unsigned long long* b = &t[g * 64 * 8];
        const int last = nk < 62 ? nk - 1 : 61;
        for (int kt = 2; kt < last; kt += (last > 24 ? 3 : 1)) {
            const unsigned long long* r = &b[kt * 8];
            printf("  %3d: %5llu %5llu %5llu %5llu %5llu %5llu %5llu | %5llu\n", kt, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5],
                   r[7] - r[6], b[(kt + 1) * 8] - r[0]);
        }
        printf("  steps 2..%d: %.1f ticks per step\n", last, (double)(b[last * 8] - b[2 * 8]) / (last - 2));
    }
    hipFree(A); hipFree(W); hipFree(out); hipFree(bias);
}

int main() {
    run<256>(15360, 3072, 1024);
    run<256>(15360, 1024, 4096);
    run<128>(15360, 3072, 1024);
    // fewer busy CUs: is the K-loop step the same when the chip draws less power / shares less L2 and fabric?
    run<256>(2048, 3072, 4096);    //  96 tiles
    run<256>(512, 3072, 4096);     //  24 tiles
    run<256>(256, 256, 4096);      //   1 tile
    return 0;
}
