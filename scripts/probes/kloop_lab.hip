// Probe: K-loop structures of the chip-filling GEMM side by side on one box, random operands: time, TF/s and a checksum of the output bits
// (all structures accumulate in the same order: the checksums must agree).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/probes/kloop_lab.hip -o scripts/probes/build/kloop_lab
#define M3R_GEMM_LAB 1   // the lab kernels of scripts/probes/lab/*.inc
#include "../../must3r_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace m3r;

static void fill_random(_Float16* d, size_t n, float scale, unsigned seed) {
    std::vector<_Float16> h(n);
    unsigned s = seed;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = (_Float16)((((s >> 8) & 0xffff) / 32768.0f - 1.0f) * scale);
    }
    (void)hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}
static unsigned long long checksum(const _Float16* d, size_t n) {
    std::vector<unsigned short> h(n);
    (void)hipMemcpy(h.data(), d, n * 2, hipMemcpyDeviceToHost);
    unsigned long long c = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) c = (c ^ h[i]) * 1099511628211ull;
    return c;
}
typedef int (*Launch)(const GemmArgs&, hipStream_t);
struct Var { const char* name; Launch plain; Launch split; };

static void run(const Var* vars, int nv, int M, int N, int K, int ws) {
    _Float16 *A, *W, *out; float* bias;
    (void)hipMalloc(&A, (size_t)M * K * 2); (void)hipMalloc(&W, (size_t)N * K * 2 * ws); (void)hipMalloc(&out, (size_t)M * N * 2); (void)hipMalloc(&bias, N * 4);
    fill_random(A, (size_t)M * K, 1.0f, 1u); fill_random(W, (size_t)N * K * ws, 0.03f, 2u); (void)hipMemset(bias, 0, N * 4);
    GemmArgs a;
    std::memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.wsplit = ws == 2 ? 2 : 0; a.batch = 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("M %d N %d K %d ws %d\n", M, N, K, ws);
    for (int round = 0; round < 2; ++round)
        for (int v = 0; v < nv; ++v) {
            Launch L = ws == 2 ? vars[v].split : vars[v].plain;
            if (!L) continue;
            (void)hipMemset(out, 0, (size_t)M * N * 2);
            int rc = 0;
            for (int rep = 0; rep < 3; ++rep) rc |= L(a, 0);
            (void)hipEventRecord(e0, 0);
            for (int rep = 0; rep < 10; ++rep) rc |= L(a, 0);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("  %-26s rc %d  %8.1f us  %7.1f TF/s  sum %016llx\n", vars[v].name, rc, ms * 100.f, 2.0 * M * N * K / (ms * 1e-4) / 1e12, checksum(out, (size_t)M * N));
            fflush(stdout);
        }
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(out); (void)hipFree(bias);
}

#define PL(f) (Launch)(f)
int main() {
    const Var vars[] = {
        {"256k", PL((launch_256k<f16_t, EPI_STORE16, 1, 256>)), PL((launch_256<f16_t, EPI_STORE16, 2, 128>))},
        {"256p nph2 sync2", PL((launch_256p<f16_t, EPI_STORE16, 1, 256, 2, 2>)), PL((launch_256p<f16_t, EPI_STORE16, 2, 128, 2, 2>))},
        {"256n one phase per K-tile", PL((launch_256n<f16_t, EPI_STORE16>)), nullptr},
        {"256w pat1", PL((launch_256w<f16_t, EPI_STORE16, 1, 256, 1>)), PL((launch_256w<f16_t, EPI_STORE16, 2, 128, 1>))},
    };
    const int nv = sizeof(vars) / sizeof(vars[0]);
    run(vars, nv, 4096, 4096, 4096, 1);   // the square shapes of the vendor yardstick / the guide's template figures (1390 / 1320 TF/s)
    run(vars, nv, 8192, 8192, 8192, 1);
    run(vars, nv, 15360, 3072, 4096, 1);
    run(vars, nv, 15360, 4096, 1024, 1);
    run(vars, nv, 15360, 3072, 2048, 2);
    run(vars, nv, 30720, 4096, 1024, 1);
    run(vars, nv, 15100, 1024, 320, 1);   // ragged rows, five K-tiles
    return 0;
}
