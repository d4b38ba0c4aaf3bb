// Probe: shader-cycle stamps inside the key-tile loop of attn_kernel<f16, 32> for one block (all 4 waves) on the update
// cross-attention shape (1 view x 768 queries x 12 heads over nk keys, split-KV) or the render shape (20 views).
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATTN_TRACE=37 -I must3r_amd/csrc scripts/probes/attn_trace.hip -o /tmp/attn_trace && /tmp/attn_trace [nviews nk nsplit]
#include "../../must3r_amd/csrc/attention.hip"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace m3r;

int main(int argc, char** argv) {
    const int nviews = argc > 1 ? atoi(argv[1]) : 1, nk = argc > 2 ? atoi(argv[2]) : 7680, nsplit = argc > 3 ? atoi(argv[3]) : 7;
    const int heads = 12, nq = 768, D = heads * 64, R = nviews * nq;
    _Float16 *q, *kv, *o; float *po, *pml; AttnView* tab;
    hipMalloc(&q, (size_t)R * D * 2); hipMalloc(&kv, (size_t)nk * 2 * D * 2); hipMalloc(&o, (size_t)R * D * 2);
    hipMalloc(&po, (size_t)nsplit * R * D * 4); hipMalloc(&pml, (size_t)nsplit * R * heads * 8); hipMalloc(&tab, sizeof(AttnView) * nviews);
    std::vector<_Float16> h((size_t)nk * 2 * D);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((x >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f); }
    hipMemcpy(kv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(q, h.data(), (size_t)R * D * 2 < h.size() * 2 ? (size_t)R * D * 2 : h.size() * 2, hipMemcpyHostToDevice);
    std::vector<AttnView> tv(nviews);
    for (int i = 0; i < nviews; ++i) tv[i] = AttnView{i * nq, nq, 0, nk, 0, 0};
    hipMemcpy(tab, tv.data(), sizeof(AttnView) * nviews, hipMemcpyHostToDevice);
    AttnArgs a;
    std::memset(&a, 0, sizeof(a));
    a.Q = q; a.K = kv; a.V = kv + D; a.O = o; a.ldq = D; a.ldk = a.ldv = 2 * D; a.ldo = D; a.heads = heads; a.views = tab; a.nviews = nviews;
    a.max_nq = nq; a.scale = 0.125f; a.q_prescaled = 1; a.nsplit = nsplit; a.part_o = po; a.part_ml = pml; a.total_q_rows = R; a.dense_rows = 1;
    const int ngrp = nviews * heads, nqb = (nq + 127) / 128, npairs = ngrp * (nsplit > 1 ? nsplit : 1);
    const int grid = ((npairs + 7) / 8) * 8 * nqb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((attn_kernel<f16_t, 32>), dim3(grid), dim3(256), 0, 0, a, nqb, ngrp, nsplit > 1 ? nsplit : 1);
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((attn_kernel<f16_t, 32>), dim3(grid), dim3(256), 0, 0, a, nqb, ngrp, nsplit > 1 ? nsplit : 1);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("nviews %d nk %d nsplit %d grid %d: %.1f us per launch (instrumented)\n", nviews, nk, nsplit, grid, ms * 100);
    std::vector<unsigned long long> t(4 * 64 * 8);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_attn_trace), t.size() * 8);
    for (int w = 0; w < 4; w += 3) {
        printf("wave %d   tile: wait_dma barrier stage_issue QK softmax PV | tile total (shader cycles)\n", w);
        for (int it = 1; it < 14; ++it) {
            const unsigned long long* r = &t[(w * 64 + it) * 8];
            if (!r[6]) break;
            printf("  %3d: %6llu %6llu %6llu %6llu %6llu %6llu | %6llu\n", it, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5],
                   t[(w * 64 + it + 1) * 8] ? t[(w * 64 + it + 1) * 8] - r[0] : r[6] - r[0]);
        }
    }
    return 0;
}
