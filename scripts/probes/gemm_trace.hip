// Probe: cycle stamps inside the K-loop of the single-round small-M GEMM (8 waves, 64x64 tile, split weights, 6-slot
// ring, fragment prefetch) for block 0 / thread 0.  Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGEMM_TRACE -I must3r_amd/csrc scripts/probes/gemm_trace.hip -o /tmp/gemm_trace && /tmp/gemm_trace
#include "../../must3r_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace m3r;

int main() {
    const int M = 768, N = 768, K = 3072;
    _Float16 *A, *W; float *x, *bias;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2 * 2); hipMalloc(&x, (size_t)M * N * 4); hipMalloc(&bias, N * 4);
    hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(W, 0x2c, (size_t)N * K * 4); hipMemset(x, 0, (size_t)M * N * 4); hipMemset(bias, 0, N * 4);
    GemmArgs a;
    std::memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.out = x; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.wsplit = 2; a.batch = 1;
    const size_t lds = (size_t)6 * (64 + 2 * 64) * 64 * 2;
    auto kern = gemm_kernel<f16_t, 64, 64, 4, 2, EPI_RESID_F32, 6, 2, 64, 1>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(144), dim3(512), lds, 0, a);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(64 * 8);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_gemm_trace), t.size() * 8);
    printf("iter: wait_dma barrier stage lgkm_touch load_frags mfma_issue | iteration total (s_memtime ticks, 100 MHz => x24 for 2.4 GHz cycles?)\n");
    for (int kt = 2; kt < 46; kt += 4) {
        const unsigned long long* r = &t[kt * 8];
        printf("%3d: %6llu %6llu %6llu %6llu %6llu %6llu | %6llu\n", kt, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5],
               t[(kt + 1) * 8] - r[0]);
    }
    unsigned long long tot = t[45 * 8] - t[2 * 8];
    printf("43 iterations: %llu ticks = %.1f per iteration\n", tot, tot / 43.0);
    return 0;
}
