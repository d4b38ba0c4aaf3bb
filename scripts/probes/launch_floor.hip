// Probe: what does a kernel boundary cost, and how much of it is the kernel-argument fetch?  Chains of dependent launches
// (same stream) of a kernel whose blocks do one dependent global load + store (the shortest critical path a real kernel has):
//   struct   : arguments in a 256-byte struct passed by value (how GemmArgs / AttnArgs travel) -> s_load from the kernarg segment
//   preload  : the same fields the first instructions need as leading scalar arguments, compiled with
//              -mllvm -amdgpu-kernarg-preload-count=12 (delivered in SGPRs at wave launch), the rest in the struct
//   empty    : no memory access at all (pure dispatch cost)
// for 256 blocks x 512 threads (the GEMM launches of a one-view update) and 768 x 256 (the split attention).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=12 scripts/probes/launch_floor.hip -o scripts/probes/build/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Args {
    const float* a;
    float* b;
    int n, m, k, lda;
    float s;
    const float* c;
    int pad[48];
};

__global__ void __launch_bounds__(512) k_struct(const Args p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p.n) p.b[i] = p.a[i] * p.s + (float)p.pad[7];
}
__global__ void __launch_bounds__(512) k_preload(const float* a, float* b, int n, float s, const Args p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i] * s + (float)p.pad[7];
}
__global__ void __launch_bounds__(512) k_empty(const Args p) {
    if (p.n < 0) p.b[0] = 1.f;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main() {
    const int N = 768 * 512;
    float *a, *b;
    CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4));
    CK(hipMemset(a, 0, N * 4)); CK(hipMemset(b, 0, N * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int shape = 0; shape < 2; ++shape) {
        const int blocks = shape ? 768 : 256, threads = shape ? 256 : 512;
        Args p{};
        p.a = a; p.b = b; p.n = blocks * threads; p.s = 1.0f;
        for (int rep = 0; rep < 2; ++rep)
            for (int var = 0; var < 3; ++var) {
                auto chain = [&](int n) {
                    for (int it = 0; it < n; ++it) {
                        // ping-pong so that every launch depends on the previous one's output
                        p.a = (it & 1) ? b : a; p.b = (it & 1) ? a : b;
                        if (var == 0) hipLaunchKernelGGL(k_struct, dim3(blocks), dim3(threads), 0, s, p);
                        else if (var == 1) hipLaunchKernelGGL(k_preload, dim3(blocks), dim3(threads), 0, s, p.a, p.b, p.n, p.s, p);
                        else hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, s, p);
                    }
                };
                // (1) plain stream launches (host launch rate may be the bound), (2) the same chain replayed from a graph (device side only)
                chain(200);
                CK(hipEventRecord(e0, s));
                chain(iters);
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                hipGraph_t g;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
                chain(500);
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < 8; ++r) CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float msg = 0;
                CK(hipEventElapsedTime(&msg, e0, e1));
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
                printf("%4d x %3d  %-8s %6.2f us per launch (stream)  %6.2f us (graph replay)\n", blocks, threads,
                       var == 0 ? "struct" : var == 1 ? "preload" : "empty", ms * 1000.f / iters, msg * 1000.f / 4000.f);
            }
    }
    return 0;
}
