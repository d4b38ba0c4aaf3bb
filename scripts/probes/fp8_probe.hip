// Hardware-semantics probe for the fp8 attention path (gfx950): operand layouts of the fp8 MFMAs, the 8-bit transposing LDS
// read and the packed fp8 conversion.  Each hypothesis is checked with random small-integer matrices against a host matmul
// (a wrong layout cannot pass by accident); the transposing read and the conversion are dumped.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/fp8_probe.hip -o scripts/probes/build/fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// e4m3fn encoding of small integers -8..8 (exact)
__host__ __device__ static unsigned char enc_e4m3(int v) {
    if (v == 0) return 0;
    unsigned char s = v < 0 ? 0x80 : 0;
    int a = v < 0 ? -v : v;
    int e = 0;
    while ((a >> (e + 1)) != 0) ++e;          // floor(log2 a)
    int m = ((a << 3) >> e) & 7;              // 3 mantissa bits
    return s | (unsigned char)(((e + 7) << 3) | m);
}

__global__ void k_mfma16x32(const unsigned char* A, const unsigned char* B, float* D) {  // A[16][32], B[32][16] (row-major bytes)
    const int l = threadIdx.x;
    unsigned long long a = 0, b = 0;
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * (l >> 4) + e;
        a |= (unsigned long long)A[(l & 15) * 32 + k] << (8 * e);
        b |= (unsigned long long)B[k * 16 + (l & 15)] << (8 * e);
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a, (long)b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

__global__ void k_mfma16x128(const unsigned char* A, const unsigned char* B, float* D, int hyp) {  // A[16][128], B[128][16]
    const int l = threadIdx.x;
    unsigned char ab[32], bb[32];
    for (int e = 0; e < 32; ++e) {
        int k = hyp == 0 ? 32 * (l >> 4) + e : (16 * (l >> 4) + (e & 15) + 64 * (e >> 4));
        ab[e] = A[(l & 15) * 128 + k];
        bb[e] = B[k * 16 + (l & 15)];
    }
    i32x8 a, b;
    memcpy(&a, ab, 32);
    memcpy(&b, bb, 32);
    f32x4 c = {0, 0, 0, 0};
    // cbsz = 0 / blgp = 0: both operands fp8 (e4m3); scales: e8m0 127 = 2^0 in byte 0 (opsel 0)
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

__global__ void k_mfma32x64(const unsigned char* A, const unsigned char* B, float* D) {  // A[32][64], B[64][32]
    const int l = threadIdx.x;
    unsigned char ab[32], bb[32];
    for (int e = 0; e < 32; ++e) {
        const int k = 32 * (l >> 5) + e;
        ab[e] = A[(l & 31) * 64 + k];
        bb[e] = B[k * 32 + (l & 31)];
    }
    i32x8 a, b;
    memcpy(&a, ab, 32);
    memcpy(&b, bb, 32);
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// LDS byte at position p holds (p & 255); lane l passes address l*8; dump the 8 bytes each lane receives
__global__ void k_tr8(unsigned char* out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[512];
    const int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) lds[i] = (unsigned char)i;
    __syncthreads();
    u32x2 r;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds + l * 8);
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
    memcpy(out + l * 8, &r, 8);
}

__global__ void k_cvt(unsigned* out) {
    const float v[8] = {1.0f, 2.0f, -0.5f, 448.0f, 1000.0f, 0.001f, 3.3f, -1000.0f};
    for (int i = 0; i < 4; ++i) {
        int o = 0;
        o = __builtin_amdgcn_cvt_pk_fp8_f32(v[2 * i], v[2 * i + 1], o, false);   // low 16 bits
        o = __builtin_amdgcn_cvt_pk_fp8_f32(v[2 * i + 1], v[2 * i], o, true);    // high 16 bits
        out[i] = (unsigned)o;
    }
}

static bool check(const char* name, const std::vector<float>& got, const std::vector<int>& A, const std::vector<int>& B, int M, int N, int K) {
    double worst = 0;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            long s = 0;
            for (int k = 0; k < K; ++k) s += (long)A[i * K + k] * B[k * N + j];
            double d = fabs((double)got[i * N + j] - (double)s);
            if (d > worst) worst = d;
        }
    printf("%-44s %s (max |diff| %.1f)\n", name, worst == 0 ? "MATCH" : "mismatch", worst);
    return worst == 0;
}

int main() {
    srand(1);
    auto run = [&](int M, int N, int K, int which, int hyp, const char* name) {
        std::vector<int> A(M * K), B(K * N);
        std::vector<unsigned char> a8(M * K), b8(K * N);
        for (auto& v : A) v = rand() % 9 - 4;
        for (auto& v : B) v = rand() % 9 - 4;
        for (int i = 0; i < M * K; ++i) a8[i] = enc_e4m3(A[i]);
        for (int i = 0; i < K * N; ++i) b8[i] = enc_e4m3(B[i]);
        unsigned char *dA, *dB; float* dD;
        hipMalloc(&dA, a8.size()); hipMalloc(&dB, b8.size()); hipMalloc(&dD, M * N * 4);
        hipMemcpy(dA, a8.data(), a8.size(), hipMemcpyHostToDevice);
        hipMemcpy(dB, b8.data(), b8.size(), hipMemcpyHostToDevice);
        if (which == 0) hipLaunchKernelGGL(k_mfma16x32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        if (which == 1) hipLaunchKernelGGL(k_mfma16x128, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp);
        if (which == 2) hipLaunchKernelGGL(k_mfma32x64, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        std::vector<float> D(M * N);
        hipMemcpy(D.data(), dD, M * N * 4, hipMemcpyDeviceToHost);
        check(name, D, A, B, M, N, K);
        hipFree(dA); hipFree(dB); hipFree(dD);
    };
    run(16, 16, 32, 0, 0, "mfma_f32_16x16x32_fp8_fp8  k=8*(l>>4)+e");
    run(16, 16, 128, 1, 0, "mfma_scale_16x16x128_f8f6f4 k=32*(l>>4)+e");
    run(16, 16, 128, 1, 1, "mfma_scale_16x16x128_f8f6f4 k=16*(l>>4)+e%16+64*(e/16)");
    run(32, 32, 64, 2, 0, "mfma_scale_32x32x64_f8f6f4  k=32*(l>>5)+e");
    unsigned char* dO; hipMalloc(&dO, 512);
    hipLaunchKernelGGL(k_tr8, dim3(1), dim3(64), 0, 0, dO);
    unsigned char o[512];
    hipMemcpy(o, dO, 512, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b8: lane -> 8 source byte positions (lane l passed address 8*l)\n");
    for (int l = 0; l < 64; ++l) {
        printf("  lane %2d:", l);
        for (int e = 0; e < 8; ++e) printf(" %3d", o[l * 8 + e]);
        printf("\n");
    }
    unsigned* dC; hipMalloc(&dC, 16);
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(1), 0, 0, dC);
    unsigned c[4];
    hipMemcpy(c, dC, 16, hipMemcpyDeviceToHost);
    printf("cvt_pk_fp8_f32 (low: v0,v1; high: v1,v0): {1,2}=%08x {-0.5,448}=%08x {1000,0.001}=%08x {3.3,-1000}=%08x\n", c[0], c[1], c[2], c[3]);
    return 0;
}
