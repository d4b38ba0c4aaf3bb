// the SLAM keyframe test's nearest-neighbour search (must3r/slam/nns.py:40-92,
// must3r/slam/model.py:62-91) on the GPU.  The reference builds a scipy KD-tree over ALL keyframe points after every
// keyframe (nns.py:47-50: O(n log n) per keyframe on the host) and queries it with 4 workers (nns.py:56); here the
// database simply stays in HBM and a query is an exact brute-force scan:
//
//   nn_query_kernel      d2[i] = min_j |q_i - db_j|^2   (fp32; 3 sub + 1 mul + 2 fma + 1 min per pair, packed two pairs wide)
//                        block = 256 threads x 4 queries; the database is streamed through LDS in tiles of 2048
//                        points repacked to float4 (one broadcast ds_read_b128 per point serves 4 x 64 x 4 pairs);
//                        the grid is (query blocks) x (database splits) so that a 12 k-point query batch still fills
//                        256 CUs; splits merge with atomicMin on the bit pattern (d2 >= 0, so unsigned order =
//                        float order; min is order-independent => deterministic)
//   nn_finish_kernel     dist = sqrt(d2)  (+inf stays +inf: empty database, nns.py:53-54)
//   quadrant_id_kernel   nns.py:80-92 / slam/tools.py:9-31: view-direction quadrant of (p - cam_center)
//
// VALU-bound (fp32 vector rate 157 TFLOP/s): 8 flops per pair.
#include "common.hpp"
#include "kernels.hpp"

namespace m3r {

constexpr int NN_T = 256;      // threads per block
constexpr int NN_QPT = 4;      // queries per thread
constexpr int NN_TILE = 2048;  // database points per LDS tile (32 KB as float4)

__global__ void __launch_bounds__(NN_T) nn_query_kernel(const float* __restrict__ db, const long long n_db, const float* __restrict__ q,
                                                        const long long n_q, unsigned* __restrict__ d2_bits, const long long chunk) {
    __shared__ f32x4 tile[NN_TILE];
    const long long q0 = (long long)blockIdx.x * (NN_T * NN_QPT);
    const long long j_begin = (long long)blockIdx.y * chunk;
    const long long j_end = j_begin + chunk < n_db ? j_begin + chunk : n_db;
    // queries in PAIRS: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 do two (query, point) pairs per instruction
    // (3 + 1 + 2 packed ops + 2 v_min_f32 per two pairs = 4 VALU per pair instead of 7)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 qx[NN_QPT / 2], qy[NN_QPT / 2], qz[NN_QPT / 2];
    float best[NN_QPT];
#pragma unroll
    for (int k = 0; k < NN_QPT; ++k) {
        long long i = q0 + (long long)k * NN_T + threadIdx.x;     // consecutive threads -> consecutive points
        i = i < n_q ? i : n_q - 1;
        qx[k >> 1][k & 1] = q[i * 3 + 0];
        qy[k >> 1][k & 1] = q[i * 3 + 1];
        qz[k >> 1][k & 1] = q[i * 3 + 2];
        best[k] = INFINITY;
    }
    auto visit = [&](const f32x4 p) {
        const f32x2 px = {p[0], p[0]}, py = {p[1], p[1]}, pz = {p[2], p[2]};
#pragma unroll
        for (int h = 0; h < NN_QPT / 2; ++h) {
            const f32x2 dx = qx[h] - px, dy = qy[h] - py, dz = qz[h] - pz;
            const f32x2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            best[2 * h] = fminf(best[2 * h], d2[0]);
            best[2 * h + 1] = fminf(best[2 * h + 1], d2[1]);
        }
    };
    for (long long j0 = j_begin; j0 < j_end; j0 += NN_TILE) {
        const int n = (int)(j_end - j0 < NN_TILE ? j_end - j0 : NN_TILE);
        __syncthreads();
        for (int t = threadIdx.x; t < n; t += NN_T) {
            const float* src = db + (j0 + t) * 3;
            tile[t] = f32x4{src[0], src[1], src[2], 0.f};
        }
        __syncthreads();
        int t = 0;
        for (; t + 4 <= n; t += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) visit(tile[t + u]);
        }
        for (; t < n; ++t) visit(tile[t]);
    }
#pragma unroll
    for (int k = 0; k < NN_QPT; ++k) {
        const long long i = q0 + (long long)k * NN_T + threadIdx.x;
        if (i < n_q && best[k] < INFINITY) atomicMin(d2_bits + i, __float_as_uint(best[k]));
    }
}

__global__ void nn_fill_kernel(unsigned* __restrict__ d2_bits, const long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d2_bits[i] = 0x7f800000u;   // +inf
}

__global__ void nn_finish_kernel(float* __restrict__ d, const long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = sqrtf(d[i]);         // sqrt(+inf) = +inf
}

// slam/tools.py:9-31 with quadrant_divider = div, eps = 1e-5 (float32 like the numpy code fed with float32 points)
__global__ void quadrant_id_kernel(const float* __restrict__ pts, const long long n, const float cx, const float cy, const float cz,
                                   const int div, int* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float eps = 1e-5f, pi = 3.14159265358979323846f;
    float x = pts[i * 3 + 0] - cx, y = pts[i * 3 + 1] - cy, z = pts[i * 3 + 2] - cz;
    const float nrm = fmaxf(sqrtf(x * x + y * y + z * z), eps);
    x /= nrm; y /= nrm; z /= nrm;
    float theta = acosf(z) / pi;
    float phi = atan2f(y, x) / pi;
    theta = fminf(fmaxf(theta, eps), 1.0f - eps);
    phi = fminf(fmaxf(phi, -1.0f + eps), 1.0f - eps);
    const int ti = (int)floorf(theta * (float)div);
    const int pj = (int)floorf(phi * (float)div) + div;
    out[i] = ti + pj * div;
}

int launch_nn_query(const float* db, long long n_db, const float* q, long long n_q, float* out_dist, hipStream_t s, const char** err) {
    if (n_q <= 0) return 0;
    unsigned* bits = reinterpret_cast<unsigned*>(out_dist);
    const unsigned gq = (unsigned)((n_q + 255) / 256);
    hipLaunchKernelGGL(nn_fill_kernel, dim3(gq), dim3(256), 0, s, bits, n_q);
    if (n_db > 0) {
        const long long qblocks = (n_q + NN_T * NN_QPT - 1) / (NN_T * NN_QPT);
        // database splits: enough blocks for ~4 per CU, at least one tile per split
        long long splits = (1024 + qblocks - 1) / qblocks;
        const long long max_splits = (n_db + NN_TILE - 1) / NN_TILE;
        splits = splits < max_splits ? splits : max_splits;
        splits = splits < 1 ? 1 : (splits > 65535 ? 65535 : splits);
        long long chunk = (n_db + splits - 1) / splits;
        chunk = ((chunk + NN_TILE - 1) / NN_TILE) * NN_TILE;
        splits = (n_db + chunk - 1) / chunk;
        if (qblocks > 0x7fffffffLL) { *err = "nn_query: too many queries"; return 1; }
        hipLaunchKernelGGL(nn_query_kernel, dim3((unsigned)qblocks, (unsigned)splits), dim3(NN_T), 0, s, db, n_db, q, n_q, bits, chunk);
    }
    hipLaunchKernelGGL(nn_finish_kernel, dim3(gq), dim3(256), 0, s, out_dist, n_q);
    if (hipGetLastError() != hipSuccess) { *err = "nn_query: launch failed"; return 1; }
    return 0;
}

int launch_quadrant_ids(const float* pts, long long n, const float* cam_center_host, int div, int* out, hipStream_t s, const char** err) {
    if (n <= 0) return 0;
    if (div < 1) { *err = "quadrant_ids: bad divider"; return 1; }
    hipLaunchKernelGGL(quadrant_id_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pts, n, cam_center_host[0], cam_center_host[1],
                       cam_center_host[2], div, out);
    if (hipGetLastError() != hipSuccess) { *err = "quadrant_ids: launch failed"; return 1; }
    return 0;
}

}  // namespace m3r
