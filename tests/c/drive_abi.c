/* Plain-C caller of libmust3r_hip (tests/test_zz_abi_gpu.py): the drop-in boundary driven the way a non-Python host would --
 * create, load_weight per state-dict key, finalize, encode, decode (memory update of B scenes, then render), bounds-check refusal.
 *
 *   weights file : repeated { int32 name_len; char name[name_len]; int32 ndim; int64 shape[ndim]; float data[prod(shape)] }
 *   images file  : float [B*V][3][H][W]
 *   output file  : float update pointmaps [B][V][H][W][7], then render pointmaps [B][V][H][W][7]
 * usage: drive_abi weights.bin images.bin out.bin B V H W dtype img_size enc_dim enc_depth enc_heads dec_dim dec_depth dec_heads
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "must3r_hip.h"

#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "%s failed: %s\n", #x, must3r_hip_last_error()); return 10; } } while (0)
#define HIP(x) do { if ((x) != hipSuccess) { fprintf(stderr, "%s failed\n", #x); return 11; } } while (0)

int main(int argc, char** argv) {
    if (argc != 16) { fprintf(stderr, "usage\n"); return 1; }
    const int B = atoi(argv[4]), V = atoi(argv[5]), H = atoi(argv[6]), W = atoi(argv[7]), dtype = atoi(argv[8]);
    must3r_hip_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.img_size = atoi(argv[9]); cfg.patch_size = 16;
    cfg.enc_dim = atoi(argv[10]); cfg.enc_depth = atoi(argv[11]); cfg.enc_heads = atoi(argv[12]);
    cfg.dec_dim = atoi(argv[13]); cfg.dec_depth = atoi(argv[14]); cfg.dec_heads = atoi(argv[15]);
    cfg.mlp_ratio = 4; cfg.rope_freq = 100.0f; cfg.rope_f0 = 1.0f;
    if (must3r_hip_abi_version() != MUST3R_HIP_ABI_VERSION) return 2;
    must3r_hip_ctx* c = NULL;
    CHECK(must3r_hip_create(&cfg, 0, &c));

    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    int32_t nl;
    while (fread(&nl, 4, 1, f) == 1) {
        char name[256];
        int32_t nd;
        int64_t shape[8], n = 1;
        if (nl <= 0 || nl >= 256 || fread(name, 1, (size_t)nl, f) != (size_t)nl) return 4;
        name[nl] = 0;
        if (fread(&nd, 4, 1, f) != 1 || nd < 0 || nd > 8 || fread(shape, 8, (size_t)nd, f) != (size_t)nd) return 4;
        for (int i = 0; i < nd; ++i) n *= shape[i];
        float* w = (float*)malloc((size_t)n * 4);
        if (fread(w, 4, (size_t)n, f) != (size_t)n) return 4;
        CHECK(must3r_hip_load_weight(c, name, w, 0, nd, shape));
        free(w);
    }
    fclose(f);
    CHECK(must3r_hip_finalize_weights(c, MUST3R_PART_ENCODER | MUST3R_PART_DECODER));

    const int N = (H / 16) * (W / 16), nv = B * V, L = cfg.dec_depth, D = cfg.dec_dim;
    const size_t img_n = (size_t)nv * 3 * H * W, tok_n = (size_t)nv * N * cfg.enc_dim, pm_n = (size_t)nv * H * W * 7;
    float* himg = (float*)malloc(img_n * 4);
    f = fopen(argv[2], "rb");
    if (!f || fread(himg, 4, img_n, f) != img_n) return 5;
    fclose(f);
    float *img, *tok, *pm_u, *pm_r;
    int64_t* pos;
    HIP(hipMalloc((void**)&img, img_n * 4)); HIP(hipMalloc((void**)&tok, tok_n * 4)); HIP(hipMalloc((void**)&pos, (size_t)nv * N * 16));
    HIP(hipMalloc((void**)&pm_u, pm_n * 4)); HIP(hipMalloc((void**)&pm_r, pm_n * 4));
    HIP(hipMemcpy(img, himg, img_n * 4, hipMemcpyHostToDevice));
    CHECK(must3r_hip_encode(c, dtype, img, nv, H, W, tok, pos, NULL));

    /* memory: per layer [B][cap][2 D] 16-bit, cap = V * N rows per scene */
    const int cap = V * N;
    void* mem[64];
    for (int l = 0; l < L; ++l) HIP(hipMalloc(&mem[l], (size_t)B * cap * 2 * D * 2));
    must3r_hip_group g;
    must3r_hip_decode_args a;
    memset(&g, 0, sizeof(g));   /* (pointmaps_scene_stride = 0: contiguous outputs) */
    memset(&a, 0, sizeof(a));
    a.dtype = dtype; a.mem_mode = MUST3R_MEM_KV; a.n_groups = 1; a.groups = &g; a.mem = mem;
    a.mem_capacity = cap; a.n_scenes = B; a.mem_scene_stride = cap;
    /* update, all V views of every scene in ONE call (tokens are [B][V][N][C] = the encoder's output order) */
    g.tokens = tok; g.pos = pos; g.n_views = V; g.n_tokens = N; g.H = H; g.W = W; g.pointmaps = pm_u;
    a.first_call = 1; a.n_mem = 0; a.render = 0;
    CHECK(must3r_hip_decode(c, &a, NULL));
    /* a second update would overrun the buffers: must be refused, nothing written */
    a.first_call = 0; a.n_mem = cap;
    if (must3r_hip_decode(c, &a, NULL) == 0) { fprintf(stderr, "overrun not refused\n"); return 6; }
    if (!strstr(must3r_hip_last_error(), "memory buffers hold")) { fprintf(stderr, "unexpected error: %s\n", must3r_hip_last_error()); return 7; }
    /* render every view against the final memory */
    g.pointmaps = pm_r; a.render = 1;
    CHECK(must3r_hip_decode(c, &a, NULL));
    HIP(hipDeviceSynchronize());
    float* out = (float*)malloc(pm_n * 4);
    f = fopen(argv[3], "wb");
    if (!f) return 8;
    HIP(hipMemcpy(out, pm_u, pm_n * 4, hipMemcpyDeviceToHost)); fwrite(out, 4, pm_n, f);
    HIP(hipMemcpy(out, pm_r, pm_n * 4, hipMemcpyDeviceToHost)); fwrite(out, 4, pm_n, f);
    fclose(f);
    must3r_hip_destroy(c);
    printf("ok\n");
    return 0;
}
