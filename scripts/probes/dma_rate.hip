// Probe: sustained per-CU rate of global_load_lds (16 B/lane LDS-DMA) vs global_load_dwordx4 (to VGPRs) from an
// L2/MALL-resident buffer, as a function of waves per block and loads in flight.  One block per CU (LDS sized to forbid two).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/dma_rate.hip -o /tmp/dma_rate && /tmp/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>   // MODE 0: global_load_lds, 1: global_load_dwordx4 -> regs
__global__ void probe(const char* __restrict__ src, size_t span_per_block, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    const char* base = src + (size_t)blockIdx.x * span_per_block;
    f32x4 acc = {0, 0, 0, 0};
    size_t off = (size_t)wave * 1024 + lane * 16;
    const size_t step = (size_t)nw * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const char* p = base + (off % span_per_block);
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(lds + ((wave * DEPTH + d) * 1024)), 16, 0, 0);
            } else {
                acc += *reinterpret_cast<const f32x4*>(p);
            }
            off += step;
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 0) acc[0] = reinterpret_cast<float*>(lds)[threadIdx.x];
    if (acc[0] == 12345.678f) sink[0] = acc[1] + acc[2] + acc[3];
}

template <int MODE, int DEPTH>
void run(const char* buf, size_t span, int waves, float* sink) {
    const int blocks = 256, iters = 2000 / DEPTH;
    const size_t lds = 100 * 1024;   // one block per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, DEPTH><<<blocks, waves * 64, lds>>>(buf, span, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE, DEPTH><<<blocks, waves * 64, lds>>>(buf, span, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * waves * 1024.0 * DEPTH * iters;
    printf("mode %s depth %2d waves %2d span %6zu KB/block: %7.1f GB/s total, %6.1f GB/s per CU = %5.1f B/clk/CU @2.4GHz\n",
           MODE == 0 ? "glds" : "vgpr", DEPTH, waves, span >> 10, bytes / ms / 1e6, bytes / ms / 1e6 / blocks, bytes / ms / 1e6 / blocks / 2.4);
}

int main() {
    const size_t total = 256ull * 4 * 1024 * 1024;
    char* buf; float* sink;
    hipMalloc(&buf, total); hipMalloc(&sink, 64);
    hipMemset(buf, 1, total);
    for (size_t span : {size_t(64) << 10, size_t(1) << 20}) {   // 64 KB/block (16 MB total: L2-resident) and 1 MB/block (256 MB: MALL/HBM)
        for (int waves : {4, 8, 16}) {
            run<0, 2>(buf, span, waves, sink);
            run<0, 6>(buf, span, waves, sink);
            run<0, 12>(buf, span, waves, sink);
            run<1, 6>(buf, span, waves, sink);
            run<1, 12>(buf, span, waves, sink);
        }
    }
    return 0;
}
