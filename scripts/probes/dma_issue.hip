// Probe: what does ONE LDS-DMA piece (1 KB, 16 B per lane) cost a wave that is streaming MFMAs -- one wave per SIMD, a piece every G MFMAs -- for the
// three instruction forms (flat 64-bit address VGPR pair | scalar base + 32-bit VGPR offset | buffer descriptor + VGPR offset + scalar offset) and for the
// same stream with ds_read_b128 in place of the DMA.  All 256 CUs run it; every wave re-reads a 64 KB window (L2 hits).  s_memtime over the loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/dma_issue.hip -o scripts/probes/build/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// FORM 0 none, 1 flat global_load_lds, 2 saddr global_load_lds (asm), 3 buffer_load lds (asm), 4 ds_read_b128 instead
template <int FORM, int G, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(256 * WAVES_PER_SIMD) probe(const char* __restrict__ src, int iters, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[8];
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); }
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const char* wbase = src + (size_t)blockIdx.x * (1 << 20) + (size_t)wave * 65536;   // this wave's 64 KB window
    const unsigned voff = (unsigned)((lane >> 3) * 512 + (lane & 7) * 16);               // 8 rows x 128 B of a 512-byte-pitch matrix
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + wave * 16384;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), 0, 65536, 0x00020000);
    f16x8 r0 = a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned soff = (unsigned)((it & 15) * 4096);
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[g & 7], 0, 0, 0);
        if constexpr (FORM == 1) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + soff + voff),
                                             (__attribute__((address_space(3))) void*)(lds + wave * 16384 + (it & 15) * 1024), 16, 0, 0);
        } else if constexpr (FORM == 2) {
            const char* sb = wbase + soff;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(ldsb + (it & 15) * 1024) : "memory", "m0");
        } else if constexpr (FORM == 3) {
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(ldsb + (it & 15) * 1024) : "memory", "m0");
        } else if constexpr (FORM == 4) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(ldsb + lane * 16 + (it & 15) * 1024));
        }
        if constexpr (FORM >= 1 && FORM <= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        if constexpr (FORM == 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    s += (float)r0[0];
    if (s == 12345.678f) sink[0] = s + reinterpret_cast<float*>(lds)[threadIdx.x];
    if (blockIdx.x == 7 && lane == 0) out[wave] = t1 - t0;
}

template <int FORM, int G, int WPS>
static void run(const char* buf, unsigned long long* dout, float* sink, const char* name) {
    const int iters = 4000;
    const size_t lds = 144 * 1024;
    auto k = probe<FORM, G, WPS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<256, 256 * WPS, lds>>>(buf, 100, dout, sink);
    (void)hipEventRecord(e0);
    k<<<256, 256 * WPS, lds>>>(buf, iters, dout, sink);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t[8] = {0};
    (void)hipMemcpy(t, dout, sizeof(t), hipMemcpyDeviceToHost);
    printf("%-34s G %2d waves/SIMD %d: %7.1f ticks per iteration (wave 0), %6.1f per MFMA; wall %7.1f ns per iteration\n", name, G, WPS, (double)t[0] / iters,
           (double)t[0] / iters / G, ms * 1e6 / iters);
}

int main() {
    char* buf; unsigned long long* dout; float* sink;
    (void)hipMalloc(&buf, 256ull << 20); (void)hipMalloc(&dout, 64); (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 1, 256ull << 20);
#define ALL(G, W) run<0, G, W>(buf, dout, sink, "MFMA only"); run<1, G, W>(buf, dout, sink, "+ flat global_load_lds"); run<2, G, W>(buf, dout, sink, "+ saddr global_load_lds"); \
    run<3, G, W>(buf, dout, sink, "+ buffer_load lds"); run<4, G, W>(buf, dout, sink, "+ ds_read_b128");
    ALL(4, 1) ALL(8, 1) ALL(2, 1) ALL(4, 2) ALL(8, 2)
    return 0;
}
