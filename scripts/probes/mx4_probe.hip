// Hardware-semantics + rate probe for the MX-fp4 low-part product of the split-weight GEMMs (gfx950):
//   * operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with e2m1 (fp4) operands: which nibble is the even k, which 32 values a lane supplies
//   * what the per-lane e8m0 scale byte multiplies (the lane's own 32-value block of its row / column) and what op_sel selects
//   * v_cvt_scalef32_pk_fp4_f16: direction of the scale, rounding, saturation, byte placement
//   * issue rate of the fp4 MFMA alone and mixed 4 : 1 with v_mfma_f32_16x16x32_f16 on the same accumulators (the stream the GEMM would run)
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/mx4_probe.hip -o scripts/probes/build/mx4_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;

static const float kGrid[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static float dec_fp4(int c) { return (c & 8 ? -1.f : 1.f) * kGrid[c & 7]; }

// A codes [16][128] (row i, k), B codes [128][16]; sa [16][4], sb [16][4]: e8m0 scale of (row / column, 32-block).
// hyp bit 0: even k in the LOW nibble (0) or the HIGH nibble (1).  The lane's scale VGPR carries its block's scale in byte `sel`.
__global__ void k_fp4(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb, float* D, int hyp, int sel) {
    const int l = threadIdx.x;
    unsigned char ab[16], bb[16];
    for (int e = 0; e < 16; ++e) {
        const int k0 = 32 * (l >> 4) + 2 * e, k1 = k0 + 1;
        const int a0 = A[(l & 15) * 128 + k0], a1 = A[(l & 15) * 128 + k1];
        const int b0 = B[k0 * 16 + (l & 15)], b1 = B[k1 * 16 + (l & 15)];
        ab[e] = (hyp & 1) ? (unsigned char)((a0 << 4) | a1) : (unsigned char)((a1 << 4) | a0);
        bb[e] = (hyp & 1) ? (unsigned char)((b0 << 4) | b1) : (unsigned char)((b1 << 4) | b0);
    }
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    memcpy(&a, ab, 16);
    memcpy(&b, bb, 16);
    // scale registers: the wanted byte at position `sel`, junk (0x55 pattern = 2^-42) elsewhere so that a wrong byte select shows
    const unsigned sva = (0x55555555u & ~(0xffu << (8 * sel))) | ((unsigned)sa[(l & 15) * 4 + (l >> 4)] << (8 * sel));
    const unsigned svb = (0x55555555u & ~(0xffu << (8 * sel))) | ((unsigned)sb[(l & 15) * 4 + (l >> 4)] << (8 * sel));
    f32x4 c = {0, 0, 0, 0};
    if (sel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, 0, (int)sva, 0, (int)svb);
    else if (sel == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, 1, (int)sva, 1, (int)svb);
    else if (sel == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, 2, (int)sva, 2, (int)svb);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, 3, (int)sva, 3, (int)svb);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// f16 pairs -> fp4 byte, with the scale operand: out[i] = {byte0 of sel 0, byte1 of sel 1, byte 2 of sel 2, byte 3 of sel 3}
__global__ void k_cvt(const _Float16* x, unsigned* out, float scale, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    h2 v = {x[2 * i], x[2 * i + 1]};
    unsigned o = 0;
    o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(o, v, scale, 0);
    o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(o, v, scale, 1);
    o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(o, v, scale, 2);
    o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(o, v, scale, 3);
    out[i] = o;
}

// rates: mode 0 = fp4 16x16x128 only, 1 = f16 16x16x32 only, 2 = per 128 k: 4 f16 + 1 fp4 on the same accumulator (8 accumulators round-robin)
template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters) {
    i32x8 a4 = {0x11111111, 0x22222222, 0x11111111, 0x22222222, 0, 0, 0, 0}, b4 = a4;
    f16x8 a16, b16;
    for (int i = 0; i < 8; ++i) { a16[i] = (_Float16)(0.001f * (threadIdx.x + i)); b16[i] = (_Float16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0 || MODE == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a4, b4, acc[i], 4, 4, 0, 127, 0, 127);
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, b16, acc[i], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    srand(3);
    const int M = 16, N = 16, K = 128;
    std::vector<unsigned char> A(M * K), B(K * N), sa(16 * 4), sb(16 * 4);
    for (auto& v : A) v = rand() & 15;
    for (auto& v : B) v = rand() & 15;
    unsigned char *dA, *dB, *dsa, *dsb; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, M * N * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int scaled = 0; scaled < 2; ++scaled)
        for (int hyp = 0; hyp < 2; ++hyp)
            for (int sel = 0; sel < (scaled ? 4 : 1); ++sel) {
                for (int i = 0; i < 64; ++i) { sa[i] = scaled ? 127 + (rand() % 7 - 3) : 127; sb[i] = scaled ? 127 + (rand() % 7 - 3) : 127; }
                hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice);
                hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(k_fp4, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD, hyp, sel);
                std::vector<float> D(M * N);
                hipMemcpy(D.data(), dD, M * N * 4, hipMemcpyDeviceToHost);
                double worst = 0;
                for (int i = 0; i < M; ++i)
                    for (int j = 0; j < N; ++j) {
                        double s = 0;
                        for (int k = 0; k < K; ++k)
                            s += (double)dec_fp4(A[i * K + k]) * ldexp(1.0, sa[i * 4 + k / 32] - 127) * dec_fp4(B[k * N + j]) * ldexp(1.0, sb[j * 4 + k / 32] - 127);
                        worst = fmax(worst, fabs(s - D[i * N + j]));
                    }
                printf("fp4 16x16x128 %s scales, even k in the %s nibble, scale byte %d: %s (max |diff| %.4f)\n", scaled ? "per-lane" : "unit", hyp ? "HIGH" : "LOW", sel,
                       worst == 0 ? "MATCH" : "mismatch", worst);
            }
    // conversion
    {
        const float vals[] = {0.f, 0.24f, 0.25f, 0.26f, 0.5f, 0.74f, 0.75f, 0.76f, 1.0f, 1.25f, 1.26f, 1.75f, 2.5f, 2.51f, 3.5f, 5.0f, 5.01f, 6.0f, 7.0f, 100.f, -0.3f, -1.3f, -6.5f, 65504.f};
        const int n = sizeof(vals) / sizeof(float) / 2;
        std::vector<_Float16> x(2 * n);
        for (int i = 0; i < 2 * n; ++i) x[i] = (_Float16)vals[i];
        _Float16* dx; unsigned* dout;
        hipMalloc(&dx, 4 * n); hipMalloc(&dout, 4 * n);
        hipMemcpy(dx, x.data(), 4 * n, hipMemcpyHostToDevice);
        for (float sc : {1.0f, 2.0f, 0.5f}) {
            hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, dx, dout, sc, n);
            std::vector<unsigned> o(n);
            hipMemcpy(o.data(), dout, 4 * n, hipMemcpyDeviceToHost);
            printf("v_cvt_scalef32_pk_fp4_f16 scale %.1f:", sc);
            for (int i = 0; i < n; ++i) {
                const int b0 = o[i] & 0xff;
                const bool same = ((o[i] >> 8) & 0xff) == (unsigned)b0 && ((o[i] >> 16) & 0xff) == (unsigned)b0 && ((o[i] >> 24) & 0xff) == (unsigned)b0;
                printf("  (%g,%g)->%02x=(%g,%g)%s", vals[2 * i], vals[2 * i + 1], b0, dec_fp4(b0 & 15), dec_fp4(b0 >> 4), same ? "" : "!bytes differ");
            }
            printf("\n");
        }
    }
    // rates
    {
        float* dout; hipMalloc(&dout, 1024 * 256 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000;
        for (int mode = 0; mode < 3; ++mode)
            for (int wps = 1; wps <= 2; ++wps) {   // waves per SIMD: 256 CUs x 4 SIMDs x wps
                const int blocks = 256 * wps;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, dout, iters);
                    if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, dout, iters);
                    if (mode == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, dout, iters);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double n4 = (mode != 1) ? 8.0 * iters : 0, n16 = (mode != 0) ? 32.0 * iters : 0;   // per wave
                const double waves = blocks * 4.0;
                const double flops = waves * (n4 * 2.0 * 16 * 16 * 128 + n16 * 2.0 * 16 * 16 * 32);
                // cycles per instruction at 2.4 GHz nominal, per SIMD (wps waves share a SIMD)
                const double cyc = ms * 1e-3 * 2.4e9 / ((n4 + n16) * wps);
                printf("rate mode %d (%s) %d wave(s)/SIMD: %.3f ms, %.0f TF/s (1 pass of the hi product = %.0f TF/s algorithmic), %.1f nominal cycles per MFMA\n", mode,
                       mode == 0 ? "fp4 16x16x128 only" : mode == 1 ? "f16 16x16x32 only" : "4 x f16 + 1 x fp4 per 128 k", wps, ms, flops / ms * 1e-9,
                       waves * n16 * 2.0 * 16 * 16 * 32 / ms * 1e-9, cyc);
            }
    }
    return 0;
}
