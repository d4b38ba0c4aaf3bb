// postprocess(compute_cam=True) -- must3r/engine/inference.py:16-48 -- as ONE launch per group of views.
//
// Per view: activation (norm_exp on ch 0:3 and 3:6, conf = 1 + exp(ch 6), tools/geometry.py:14-18), focal by the
// Weiszfeld / IRLS scheme of dust3r's estimate_focal_knowing_depth (closed-form L2 start + 10 re-weighted steps), and the
// weighted rigid registration pts3d_local -> pts3d of roma.rigid_points_registration (weights conf - 1), assembled
// into c2w.  oracle/cam_ref.py states both third-party algorithms.
//
// HBM-bound: 28 B read + 28 B written per pixel, once.  The 11 global reductions of the focal iteration are what the
// reference pays 11 full passes over H*W for; here a view is spread over G persistent blocks (cooperative launch, all
// co-resident), every block keeps the (x/z, y/z) of its <= 16 x 1024 pixels in LDS (128 KB), and an iteration costs one
// block reduction + one per-view barrier (arrive counter + partials through agent-scope atomics) + a fixed-order sum of G partials
// (deterministic; no floating-point atomics).  All sums are accumulated in fp64 -- the reference's are fp32 -- the
// per-pixel terms are the reference's fp32 expressions.  The 3x3 Procrustes problem is solved by one thread per view
// with Horn's quaternion form (largest eigenvector of a 4x4 symmetric matrix, cyclic Jacobi in fp64), which equals
// U diag(1,1,det(UV^T)) V^T of the SVD form without the sign case analysis.
#include "common.hpp"
#include "kernels.hpp"

namespace m3r {

constexpr int CAM_T = 1024;        // threads per block
constexpr int CAM_PPT = 16;        // pixels cached per thread (upper bound; CAM_T * CAM_PPT * 8 B = 128 KB of LDS)
constexpr int CAM_NSUM = 18;       // sum w, w*x[3], w*y[3], w*y*x^T[9], sum a, sum b
constexpr int CAM_ITERS = 10;
typedef double dbl2 __attribute__((ext_vector_type(2)));

struct CamArgs {
    const float* pm;      // [V, P, 7]
    float* p3;            // [V, P, 3]
    float* pl;            // [V, P, 3]
    float* cf;            // [V, P]
    double* sums;         // [V] final focal (input of cam_solve_kernel, which also reads buffer 0 of `partial`)
    double* partial;      // [V][G][16] per-block partial registration sums (summed by cam_solve_kernel)
    dbl2* slots;       // [V][3][G] pair exchange slots, all-ones before the launch
    int V, G, H, W, ppt;
    int linear;           // ActivationType.LINEAR (head.py:13-21): channels 0:3 / 3:6 pass through; conf = 1 + exp(ch 6) either way
};

__device__ __forceinline__ void norm_exp3c(const float* v, float* o) {
    const float d = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float sc = expm1f(d) / fmaxf(d, 1e-8f);
    o[0] = v[0] * sc;
    o[1] = v[1] * sc;
    o[2] = v[2] * sc;
}

// sum over the block of n (<= CAM_NSUM) doubles per thread; result valid in red[0..n) for every thread after return
template <int N>
__device__ __forceinline__ void block_sum(double* vals, double* red /* [16][N] */, double* out /* [N] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double v = vals[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave * N + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < CAM_T / 64; ++w) s += red[w * N + threadIdx.x];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

// Exchange of one (a, b) pair per block per step between the G blocks of a view, without a counter and without fences.
// Block g owns slot g of the step's buffer (3 buffers, step n uses n % 3): it publishes its pair with ONE 16-byte
// L2-bypassing store (sc1: agent-coherent across the 8 XCDs); thread t of every block polls slot t until both halves
// differ from the all-ones sentinel the host memset left there, so all G slots are polled in parallel and a torn store
// cannot be mistaken for data.  After consuming step n a block re-arms its slot of buffer (n + 2) % 3 = buffer of step
// n - 1, which every block has finished reading (they all published step n); it is written again at step n + 2, two
// __syncthreads (vmcnt(0)) later.  The fenced alternative -- agent-scope release/acquire around an arrive counter --
// costs a full L2 write-back / invalidate scan per wave per step on gfx950 and measured 20x the data pass.
// The cooperative launch guarantees the peers are resident; the spin bound only turns a broken guarantee (e.g. a
// CU-masked stream) into wrong numbers instead of a hung device.
constexpr long long CAM_SENTINEL = -1LL;

__device__ __forceinline__ void slot_store(dbl2* p, dbl2 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ dbl2 slot_load(const dbl2* p) {
    dbl2 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// publish this block's pair for `step`, collect everybody's, sum in block order (deterministic); result in out[0..1]
__device__ __forceinline__ void exchange_pair(dbl2* slots /* [3][G] of the view */, int G, int g, int step, const double* mine,
                                              dbl2* stage /* [G] */, double* out) {
    dbl2* buf = slots + (size_t)(step % 3) * G;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-arm store of this slot (two steps ago) has landed
        slot_store(buf + g, dbl2{mine[0], mine[1]});
    }
    for (int idx = threadIdx.x; idx < G; idx += CAM_T) {
        dbl2 d = slot_load(buf + idx);
        for (unsigned spins = 0; (__double_as_longlong(d.x) == CAM_SENTINEL || __double_as_longlong(d.y) == CAM_SENTINEL) && spins < (1u << 22); ++spins) {
            __builtin_amdgcn_s_sleep(1);
            d = slot_load(buf + idx);
        }
        stage[idx] = d;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        double s = 0.0;
        for (int b = 0; b < G; ++b) s += threadIdx.x ? stage[b].y : stage[b].x;
        out[threadIdx.x] = s;
    }
    if (threadIdx.x == 0 && step >= 1)   // re-arm the slot of step - 1 for step + 2
        slot_store(slots + (size_t)((step + 2) % 3) * G + g, dbl2{__longlong_as_double(CAM_SENTINEL), __longlong_as_double(CAM_SENTINEL)});
    __syncthreads();
}

__device__ __forceinline__ void put(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One block (64 threads) per view: lanes 0..15 add up the G per-block partials of the registration sums in block
// order, lane 0 solves.  S = [sum w, sum w x (3), sum w y (3), sum w y_i x_j (9)]
__global__ void __launch_bounds__(64) cam_solve_kernel(const double* __restrict__ partial, const double* __restrict__ focal_in, int G,
                                                       float* __restrict__ focal_out, float* __restrict__ c2w_out) {
    __shared__ double S4[4][16];
    __shared__ double S[16];
    const int v = blockIdx.x;
    const double* part = partial + (size_t)v * G * 16;
    {   // lane (q, j): blocks b = q, q + 4, ... of sum j; the four quarter sums are then added in order
        const int j = threadIdx.x & 15, q = threadIdx.x >> 4;
        double acc = 0.0;
        for (int b = q; b < G; b += 4) acc += part[(size_t)b * 16 + j];
        S4[q][j] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 16) S[threadIdx.x] = ((S4[0][threadIdx.x] + S4[1][threadIdx.x]) + S4[2][threadIdx.x]) + S4[3][threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;
    float* c2w = c2w_out + (size_t)v * 16;
    const double sw = S[0];
    double xb[3], yb[3], M[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        xb[i] = S[1 + i] / sw;
        yb[i] = S[4 + i] / sw;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i][j] = S[7 + i * 3 + j] / sw - yb[i] * xb[j];
    // Horn: s[a][b] = sum x_a y_b = M^T
    const double sxx = M[0][0], sxy = M[1][0], sxz = M[2][0];
    const double syx = M[0][1], syy = M[1][1], syz = M[2][1];
    const double szx = M[0][2], szy = M[1][2], szz = M[2][2];
    double A[4][4] = {{sxx + syy + szz, syz - szy, szx - sxz, sxy - syx},
                      {syz - szy, sxx - syy - szz, sxy + syx, szx + sxz},
                      {szx - sxz, sxy + syx, -sxx + syy - szz, syz + szy},
                      {sxy - syx, szx + sxz, syz + szy, -sxx - syy + szz}};
    double Q[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    double scale = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) scale = fmax(scale, fabs(A[i][j]));
    const double inv_scale = scale > 0.0 ? 1.0 / scale : 1.0;   // eigenvectors do not care; keeps the fp32 angle in range
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) A[i][j] *= inv_scale;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) off = fmax(off, fabs(A[p][q]));
        if (!(off > 1e-14)) break;   // R is returned in fp32
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double apq = A[p][q];
                if (fabs(apq) > 1e-30) {
                    // the angle only has to be roughly right (fp32 tangent); what must hold to fp64 is c^2 + s^2 = 1, so
                    // c comes from a Newton-refined fp64 rsqrt of 1 + t^2 and s = t c.  fp64 div/sqrt cost ~30
                    // dependent instructions each and this is a single thread per view.
                    const float thf = (float)(A[q][q] - A[p][p]) / (2.0f * (float)apq);
                    const float tf = (thf >= 0.f ? 1.0f : -1.0f) / (fabsf(thf) + sqrtf(thf * thf + 1.0f));
                    const double t = (double)tf, n2 = t * t + 1.0;
                    double c = __builtin_amdgcn_rsq(n2);
                    c = c * (1.5 - 0.5 * n2 * c * c);
                    c = c * (1.5 - 0.5 * n2 * c * c);
                    const double s = t * c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {   // A <- A J
                        const double akp = A[k][p], akq = A[k][q];
                        A[k][p] = c * akp - s * akq;
                        A[k][q] = s * akp + c * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {   // A <- J^T A
                        const double apk = A[p][k], aqk = A[q][k];
                        A[p][k] = c * apk - s * aqk;
                        A[q][k] = s * apk + c * aqk;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {   // Q <- Q J
                        const double qkp = Q[k][p], qkq = Q[k][q];
                        Q[k][p] = c * qkp - s * qkq;
                        Q[k][q] = s * qkp + c * qkq;
                    }
                }
            }
    }
    double bestv = A[0][0], qw = Q[0][0], qx = Q[1][0], qy = Q[2][0], qz = Q[3][0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (A[i][i] > bestv) {
            bestv = A[i][i];
            qw = Q[0][i]; qx = Q[1][i]; qy = Q[2][i]; qz = Q[3][i];
        }
    const double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= qn; qx /= qn; qy /= qn; qz /= qn;
    const double R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)},
                            {2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)},
                            {2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)}};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double t = yb[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            c2w[i * 4 + j] = (float)R[i][j];
            t -= R[i][j] * xb[j];
        }
        c2w[i * 4 + 3] = (float)t;
    }
    c2w[12] = 0.f; c2w[13] = 0.f; c2w[14] = 0.f; c2w[15] = 1.f;
    focal_out[v] = (float)focal_in[v];
}

__global__ void __launch_bounds__(CAM_T) cam_kernel(const CamArgs p) {
    __shared__ float2 qs[CAM_PPT * CAM_T];   // (x/z, y/z) of this block's pixels: slot k * CAM_T + thread
    __shared__ double red[(CAM_T / 64) * CAM_NSUM];
    __shared__ double tot[CAM_NSUM];
    __shared__ dbl2 stage[512];           // exchange_pair staging (G <= 512)
    const int v = blockIdx.x / p.G, g = blockIdx.x - v * p.G;
    const int P = p.H * p.W;
    const size_t base = (size_t)v * P;
    const float ppx = 0.5f * (float)p.W, ppy = 0.5f * (float)p.H;
    const int stride = p.G * CAM_T;

    double acc[CAM_NSUM];
#pragma unroll
    for (int k = 0; k < CAM_NSUM; ++k) acc[k] = 0.0;
    for (int k = 0; k < p.ppt; ++k) {
        const int i = g * CAM_T + threadIdx.x + k * stride;
        if (i < P) {
            const float* src = p.pm + (base + i) * 7;
            float r[7];
#pragma unroll
            for (int c = 0; c < 7; ++c) r[c] = src[c];
            float y[3], x[3];
            if (p.linear) {            // block-uniform
#pragma unroll
                for (int k = 0; k < 3; ++k) { y[k] = r[k]; x[k] = r[3 + k]; }
            } else {
                norm_exp3c(r, y);          // pts3d (world)
                norm_exp3c(r + 3, x);      // pts3d_local (camera)
            }
            const float conf = 1.0f + expf(r[6]);
            float* o3 = p.p3 + (base + i) * 3;
            float* ol = p.pl + (base + i) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                o3[c] = y[c];
                ol[c] = x[c];
            }
            p.cf[base + i] = conf;
            const float w = conf - 1.0f;           // the reference's weight: (1 + e^v) - 1 in fp32
            float a = x[0] / x[2], b = x[1] / x[2];
            a = (fabsf(a) <= 3.402823466e38f) ? a : 0.f;   // nan_to_num(nan=0, posinf=0, neginf=0)
            b = (fabsf(b) <= 3.402823466e38f) ? b : 0.f;
            qs[k * CAM_T + threadIdx.x] = make_float2(a, b);
            const float u = (float)(i % p.W) - ppx, vv = (float)(i / p.W) - ppy;
            acc[0] += (double)w;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[1 + c] += (double)(w * x[c]);
                acc[4 + c] += (double)(w * y[c]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) acc[7 + c * 3 + d] += (double)(w * y[c]) * (double)x[d];
            acc[16] += (double)(a * u + b * vv);
            acc[17] += (double)(a * a + b * b);
        }
    }
    double* part = p.partial + ((size_t)v * p.G + g) * 16;
    dbl2* slots = p.slots + (size_t)v * 3 * p.G;
    block_sum<CAM_NSUM>(acc, red, tot);
    if (threadIdx.x < 16) put(part + threadIdx.x, tot[threadIdx.x]);
    __shared__ double pair[2];
    exchange_pair(slots, p.G, g, 0, tot + 16, stage, pair);
    float focal = (float)(pair[0] / pair[1]);   // mean(a)/mean(b): the 1/P cancels

    // pixel (column, row) of slot k advances by a fixed (dcol, drow) -- no integer division in the iteration loop
    const int i0 = g * CAM_T + threadIdx.x;
    const int col0 = i0 % p.W, row0 = i0 / p.W, dcol = stride % p.W, drow = stride / p.W;
    for (int it = 0; it < CAM_ITERS; ++it) {
        // per-thread partial sums of <= 16 terms in fp32, everything across threads / blocks in fp64
        float sa = 0.f, sb = 0.f;
        int col = col0, row = row0, i = i0;
        for (int k = 0; k < p.ppt; ++k) {
            if (i < P) {
                const float2 q = qs[k * CAM_T + threadIdx.x];
                const float u = (float)col - ppx, vv = (float)row - ppy;
                const float dx = u - focal * q.x, dy = vv - focal * q.y;
                // w = 1 / max(|d|, 1e-8) = rsqrt(max(|d|^2, 1e-16))
                const float w = __builtin_amdgcn_rsqf(fmaxf(dx * dx + dy * dy, 1e-16f));
                sa += w * (q.x * u + q.y * vv);
                sb += w * (q.x * q.x + q.y * q.y);
            }
            i += stride;
            col += dcol;
            row += drow;
            if (col >= p.W) {
                col -= p.W;
                ++row;
            }
        }
        double it_acc[2] = {(double)sa, (double)sb};
        block_sum<2>(it_acc, red, tot);
        exchange_pair(slots, p.G, g, it + 1, tot, stage, pair);
        focal = (float)(pair[0] / pair[1]);
    }
    if (g == 0 && threadIdx.x == 0) p.sums[v] = (double)focal;
}

size_t cam_scratch_bytes(int n_views, int H, int W) {
    const int P = H * W;
    const int gmax = (P + CAM_T - 1) / CAM_T;
    // partials sized for the largest G any launch can pick, + one counter per view
    // [V][3][gmax] slots (16 B) + [V][gmax][16] partial sums + [V] focal
    return (size_t)n_views * gmax * (3 * 16 + 16 * 8) + (size_t)n_views * 8 + 256;
}

int launch_postprocess_cam(const float* pm, int linear, int n_views, int H, int W, float* p3, float* pl, float* cf, float* focal,
                           float* c2w, void* scratch, size_t scratch_bytes, hipStream_t s, const char** err) {
    if (n_views <= 0) return 0;
    const int P = H * W;
    if (P <= 0) { *err = "postprocess_cam: empty image"; return 1; }
    if (scratch_bytes < cam_scratch_bytes(n_views, H, W)) { *err = "postprocess_cam: scratch too small"; return 1; }
    int dev = 0, ncu = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cam_kernel, CAM_T, 0) != hipSuccess || per_cu < 1) {
        *err = "postprocess_cam: occupancy query failed";
        return 1;
    }
    const int slots = ncu * per_cu;                                   // blocks that can be resident together
    const int gmin = (P + CAM_T * CAM_PPT - 1) / (CAM_T * CAM_PPT);   // fewest blocks per view (register cache bound)
    const int gmax = (P + CAM_T - 1) / CAM_T;
    if (gmin > slots) { *err = "postprocess_cam: image too large for the per-view register cache"; return 1; }
    const int vmax = slots / gmin;                                    // views per launch
    // scratch: [slots | partial | focal]; the slots must read all-ones (the "not yet published" sentinel)
    const size_t slot_bytes = (size_t)n_views * 3 * gmax * 16;
    dbl2* slot_base = reinterpret_cast<dbl2*>(scratch);
    double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + slot_bytes);
    double* sums = partial + (size_t)n_views * gmax * 16;
    if (hipMemsetAsync(slot_base, 0xFF, slot_bytes, s) != hipSuccess) { *err = "postprocess_cam: memset failed"; return 1; }
    for (int v0 = 0; v0 < n_views; v0 += vmax) {
        const int nv = (n_views - v0 < vmax) ? n_views - v0 : vmax;
        int G = slots / nv;
        G = G > gmax ? gmax : G;
        // more than ~64 blocks per view only lengthens every exchange step (G slots to poll and add): measured 78 us
        // at G = 192 vs 66 us at G = 64 for one 384x512 view
        const int gcap = gmin > 64 ? gmin : 64;
        G = G > gcap ? gcap : G;
        CamArgs a;
        a.pm = pm + (size_t)v0 * P * 7;
        a.p3 = p3 + (size_t)v0 * P * 3;
        a.pl = pl + (size_t)v0 * P * 3;
        a.cf = cf + (size_t)v0 * P;
        a.sums = sums + v0;
        a.partial = partial + (size_t)v0 * gmax * 16;
        a.slots = slot_base + (size_t)v0 * 3 * gmax;
        a.V = nv; a.G = G; a.H = H; a.W = W; a.linear = linear ? 1 : 0;
        a.ppt = (P + G * CAM_T - 1) / (G * CAM_T);
        void* args[] = {&a};
        if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(cam_kernel), dim3(nv * G), dim3(CAM_T), args, 0, s) != hipSuccess) {
            (void)hipGetLastError();
            *err = "postprocess_cam: cooperative launch failed";
            return 1;
        }
        hipLaunchKernelGGL(cam_solve_kernel, dim3(nv), dim3(64), 0, s, a.partial, a.sums, G, focal + v0, c2w + (size_t)v0 * 16);
        if (hipGetLastError() != hipSuccess) { *err = "postprocess_cam: solve launch failed"; return 1; }
    }
    return 0;
}

}  // namespace m3r
