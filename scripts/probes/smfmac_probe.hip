// Probe: v_smfmac_f32_16x16x64_f16 on gfx950 -- (1) where does compressed element e of the sparse A operand's lane (row i, k-group g) land in K-slot space,
// as a function of its 2-bit index, relative to the dense B operand's (k-group g', element m') slots; (2) issue rate against the dense 16x16x32 MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/smfmac_probe.hip -o scripts/probes/build/smfmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) _Float16 h16;
typedef __attribute__((ext_vector_type(4))) float f4;

// (1) one wave.  A: lane (i = l & 15, g = l >> 4) element e holds the value 1 + e + 8 g (exact in fp16, identifies (g, e); rows all equal).
// For every (g', m'): B one-hot at lanes of group g', element m' -> D[i][j] = sum of the A values whose slot is (g', m').  out[(g' * 16 + m') * 64 + lane] = D regs r = 0.
__global__ void sem(unsigned idx, int abid_sel, float* out) {
    const int l = threadIdx.x, g = l >> 4;
    h8 a;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)(1 + e + 8 * g);
    for (int gp = 0; gp < 4; ++gp)
        for (int mp = 0; mp < 16; ++mp) {
            h16 b;
            for (int m = 0; m < 16; ++m) b[m] = (_Float16)((g == gp && m == mp) ? 1.0f : 0.0f);
            f4 c = {0, 0, 0, 0};
            f4 d;
            if (abid_sel == 0) d = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, (int)idx, 0, 0);
            else d = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, (int)idx, 0, 1);
            out[(gp * 16 + mp) * 64 + l] = d[0];
        }
}

template <int SPARSE>
__global__ void __launch_bounds__(256) rate(int iters, float* sink, unsigned long long* ticks) {
    const int l = threadIdx.x & 63;
    h8 a; h16 b;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)(0.01f * ((l * 7 + e * 3) % 17 - 8));
    for (int e = 0; e < 16; ++e) b[e] = (_Float16)(0.02f * ((l * 5 + e) % 13 - 6));
    h8 b8;
    for (int e = 0; e < 8; ++e) b8[e] = b[e];
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (SPARSE) acc[i] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, acc[i], 0x4444, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b8, acc[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
    float* out; unsigned long long* ticks; float* sink;
    (void)hipMalloc(&out, 64 * 64 * 4); (void)hipMalloc(&ticks, 64); (void)hipMalloc(&sink, 64);
    std::vector<float> h(64 * 64);
    // index patterns: every 4-slot group of a lane has two 2-bit fields; pattern p: all groups the same byte-nibble
    const unsigned pats[][2] = {{0x4444u, 0}, {0xeeeeu, 0}, {0x8888u, 0}, {0xdddd4444u, 0}, {0xdddd4444u, 1}, {0xe4e4u, 0}, {0x4e4eu, 0}};
    for (auto& pt : pats) {
        sem<<<1, 64>>>(pt[0], (int)pt[1], out);
        (void)hipMemcpy(h.data(), out, 64 * 64 * 4, hipMemcpyDeviceToHost);
        printf("idx 0x%08x abid %u: B slot (g', m') <- A values (v = 1 + e + 8 g) seen by row 0 [lane 0] / row 5 [lane 5]\n", pt[0], pt[1]);
        for (int gp = 0; gp < 4; ++gp) {
            printf("  g'=%d:", gp);
            for (int mp = 0; mp < 16; ++mp) printf(" %2.0f", h[(gp * 16 + mp) * 64 + 0]);
            printf("   |");
            for (int mp = 0; mp < 16; ++mp) printf(" %2.0f", h[(gp * 16 + mp) * 64 + 5]);
            printf("\n");
        }
    }
    for (int rep = 0; rep < 2; ++rep) {
        for (int sp = 0; sp < 2; ++sp) {
            const int iters = 20000;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            if (sp) rate<1><<<1024, 256>>>(100, sink, ticks); else rate<0><<<1024, 256>>>(100, sink, ticks);
            (void)hipEventRecord(e0);
            if (sp) rate<1><<<1024, 256>>>(iters, sink, ticks); else rate<0><<<1024, 256>>>(iters, sink, ticks);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t = 0;
            (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            const double n = 1024.0 * 4 * iters * 8;   // instructions
            const double flop = n * 16 * 16 * (sp ? 64 : 32) * 2;   // dense-equivalent
            printf("%s: %.3f ms, %.1f ns per instruction per wave-slot (4 waves per CU x 4 blocks per CU), %.0f TF/s dense-equivalent, %.1f ticks per instruction (wave 0)\n",
                   sp ? "smfmac 16x16x64 (2:4 sparse A)" : "mfma   16x16x32 (dense)      ", ms, ms * 1e6 / (iters * 8.0), flop / (ms * 1e-3) / 1e12, (double)t / (iters * 8.0));
        }
    }
    return 0;
}
