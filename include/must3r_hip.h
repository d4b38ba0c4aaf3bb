/*
 * libmust3r_hip -- C ABI of the MI355X-native MUSt3R multi-view forward path.
 *
 * Drop-in boundary: everything behind the reference's two nn.Module forwards
 *     Dust3rEncoder.forward(img, true_shape) -> (x, pos)              must3r/model/encoder.py:46-52
 *     MUSt3R.forward / forward_list(x, pos, true_shape, mem, render)  must3r/model/decoder.py:158-350
 * plus the fp32 output activation of engine/inference.py:16-27.
 *
 * Conventions
 *   - plain C types only: raw DEVICE pointers (tensor.data_ptr()), sizes, an opaque context handle and a
 *     hipStream_t passed as void*.  No torch types, no exceptions across the boundary.
 *   - every entry point returns 0 on success, non-zero on error; must3r_hip_last_error() returns a
 *     thread-local, NUL-terminated description of the last failure.
 *   - the caller owns all input/output buffers.  The context owns its weights (fp32 master + packed 16-bit
 *     copies) and a grow-only workspace arena; no allocation happens on the hot path once shapes have been
 *     seen.  A context is bound to one device and is not re-entrant (one forward in flight per context),
 *     like the reference (slam/slam.py:533 runs forwards from a single worker thread).
 *   - "16-bit" buffers hold bf16 or fp16 elements according to the `dtype` argument.
 */
#ifndef MUST3R_HIP_H
#define MUST3R_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUST3R_HIP_ABI_VERSION 8

typedef struct must3r_hip_ctx must3r_hip_ctx;

/* MFMA operand type (accumulation, softmax, LayerNorm and the residual stream are always fp32) */
/* MUST3R_F16_W2: fp16 operands with every weight matrix split as W_hi + W_lo (two MFMA passes per GEMM, fp32
 * accumulation): removes the weight-rounding term that dominates the fp16 error (precision). Only valid
 * for must3r_hip_encode / must3r_hip_decode; buffers are fp16. */
/* MUST3R_F16_WA: as MUST3R_F16_W2 for the attention-side Linears (qkv, proj, projq, projk, projv, cross proj), the patch / enc->dec
 * embeddings and the head, but the Mlp weights (fc1, fc2, feedback Mlp) are plain fp16: 2/3 of the GEMM FLOPs run ONE MFMA pass.
 * Measured / emulated error: between the two; inside the 1e-3 target. */
enum { MUST3R_BF16 = 0, MUST3R_F16 = 1, MUST3R_F16_W2 = 2, MUST3R_F16_WA = 3 };
/* OR-able flag on `dtype` (BASELINE.json configs[4], "fp8 MFMA attention path"; replaces the attention back ends of
 * must3r/model/blocks/attention.py:57-79): Q and K enter the score product as OCP e4m3 bytes through the MX-scaled
 * v_mfma_scale_f32_32x32x64_f8f6f4 (twice the 16-bit MFMA rate); the softmax, its numerators P, V, the accumulators and the outputs
 * stay fp32 / 16-bit (3 mantissa bits on V alone cost 9e-3 of pointmap error, on Q and K 2e-3: DESIGN.md section 4).
 *   must3r_hip_op_attention: Q and K ARE e4m3 arrays (ldq, ldk in bytes); V and O 16-bit (ldv, ldo in elements).
 *   must3r_hip_encode / must3r_hip_decode: q and k are quantised on the fly; with MUST3R_MEM_KV the memory buffers hold rows of
 *   [K e4m3: dec_dim bytes | V 16-bit: 2*dec_dim bytes] = 3*dec_dim bytes (3/4 of the 16-bit footprint).
 * r06: PARKED.  Measured on BASELINE.json configs[4]: +0.4 % (680.4 vs 677.9 views/s) at 1.2e-3 ... 1.4e-3 from the 16-bit path -- outside the 1e-3 target for
 * nothing.  The flag is honoured only by libraries built with `make EXTRA=-DM3R_ATTN_FP8` (must3r_hip_has_fp8_attention() == 1); the default
 * library refuses it with status 1 and an error string that says so, and contains neither attn4_kernel<.., F8> nor the quantisation kernel. */
#define MUST3R_ATTN_FP8 0x100
int must3r_hip_has_fp8_attention(void);

/* layout of the caller-visible memory tensors; CachedDecoderBlock MEMORY_MODES, must3r/model/blocks/layers.py:9 */
enum { MUST3R_MEM_KV = 0, MUST3R_MEM_NORM_Y = 1, MUST3R_MEM_RAW = 2 };

/* constructor arguments of Dust3rEncoder (encoder.py:14-23) and MUSt3R (decoder.py:19-37) */
typedef struct must3r_hip_config {
    int32_t img_size, patch_size;
    int32_t enc_dim, enc_depth, enc_heads;
    int32_t dec_dim, dec_depth, dec_heads;
    int32_t mlp_ratio;
    float rope_freq; /* 'RoPE100' -> 100 (blocks/pos_embed.py:20) */
    float rope_f0;   /* F0 = old/new size (blocks/pos_embed.py:12-19) */
} must3r_hip_config;

int must3r_hip_abi_version(void);
const char* must3r_hip_last_error(void);
/* ABI 8.  Process-wide A/B switches of the library (measuring instruments, not model semantics: DESIGN.md section 10 lists them -- "PERSIST", "GEMM256",
 * "G256K", "G256P", "G256P_SPLIT", "SPARSE_256", "SPARSE_LO", "BK128", "LN_ROWS", "LNFOLD", "ENC_CHUNK_ROWS", "ATTN_LZ").  Each has a default and an allowed
 * range; an unknown name or a value outside the range is refused (status 1, must3r_hip_last_error() says why).  Without a call a switch takes its value from the
 * environment variable M3R_<NAME> (validated alike; a bad value is reported on stderr and ignored).  No counterpart in the reference: its only back-end
 * switch is toggle_memory_efficient_attention (must3r/model/blocks/attention.py:18-27). */
int must3r_hip_set_option(const char* name, long long value);

/* lifetime.  replaces: eval(encoder_args)/eval(decoder_args) + .to(device) in load_model, model/__init__.py:38-46 */
int must3r_hip_create(const must3r_hip_config* cfg, int device, must3r_hip_ctx** out);
void must3r_hip_destroy(must3r_hip_ctx* ctx);

/* Weight ingestion.  replaces: load_state_dict(strict=True), model/__init__.py:43-44.
 * `name` is the reference state-dict key prefixed by "encoder." or "decoder." (SURVEY.md section 8b), e.g.
 * "encoder.blocks_enc.0.attn.qkv.weight".  `data` is fp32, contiguous, on host (is_device=0) or device.
 * The tensor is copied.  Unknown names and shape mismatches are errors. */
int must3r_hip_load_weight(must3r_hip_ctx* ctx, const char* name, const float* data, int is_device,
                           int ndim, const int64_t* shape);
/* strict check that every parameter of the selected module(s) has been loaded; builds the fused / permuted
 * device copies (K|V projection, pixel-shuffled head, RoPE table).  The reference returns encoder and decoder
 * as two independent nn.Modules (model/__init__.py:50), so each half can be finalized on its own context. */
enum { MUST3R_PART_ENCODER = 1, MUST3R_PART_DECODER = 2 };
int must3r_hip_finalize_weights(must3r_hip_ctx* ctx, int parts);

/* Dust3rEncoder.forward (encoder.py:46-52): img fp32 [n_views,3,H,W] (one aspect ratio per call)
 *   -> tokens fp32 [n_views, N, enc_dim] (after norm_enc), pos int64 [n_views, N, 2] = (y, x); N = H/16 * W/16. */
int must3r_hip_encode(must3r_hip_ctx* ctx, int dtype, const float* img, int n_views, int H, int W,
                      float* out_tokens, int64_t* out_pos, void* stream);

/* one aspect-ratio group of a decoder call (one list entry of MUSt3R.forward_list, decoder.py:158).
 * With n_scenes = B > 1 every array carries the batch dimension in front, like the reference's tensors (decoder.py:170-186):
 * tokens [B, n_views, n_tokens, enc_dim], pos [B, n_views, n_tokens, 2], pointmaps [B, n_views, H, W, 7], all contiguous. */
typedef struct must3r_hip_group {
    const float* tokens;  /* fp32 [n_views, n_tokens, enc_dim] encoder output */
    const int64_t* pos;   /* int64 [n_views, n_tokens, 2] */
    int32_t n_views, n_tokens, H, W;
    float* pointmaps;     /* out fp32 [n_views, H, W, 7] raw head output (decoder.py:149-156) */
    /* ABI 7: elements between the pointmaps of consecutive SCENES of this group; 0 = n_views*H*W*7 (the contiguous [B, n_views, H, W, 7] of the
     * reference).  A caller that walks a scene's views over several calls (the sequential memory update) can hand every call the slice
     * [:, i:i+n] of ONE [B, V, H, W, 7] buffer instead of concatenating the calls' outputs afterwards (must3r_amd.engine.run_scenes). */
    int64_t pointmaps_scene_stride;   /* must be a multiple of 4 (the head epilogue stores 16-byte vectors); the call is refused otherwise */
} must3r_hip_group;

/* ---- ABI 8: context-parallel cross attention (SURVEY.md section 8f "later"; the keys of must3r/model/decoder.py:301-321's cross attention spread over processes) ----
 * The memory of ONE scene is SHARDED over `world` ranks (one process per GPU): every rank holds some of the memory rows of every layer, runs the same one-view
 * memory update on the same tokens (the projections / self attention / Mlp of a 768-row call are replicated, they do not shard), but attends only ITS rows and
 * contributes one PARTIAL per layer: un-normalised O fp32 [rows][dec_dim] (or, with partial16, O / l in the 16-bit operand type) followed by fp32 (m, l)
 * [rows][heads][2] -- the flash-attention partial.  The library then
 * calls `exchange`, which must leave rank r's partial in slot r on every rank (an all-gather over xGMI: RCCL's ncclAllGather, or
 * torch.distributed.all_gather_into_tensor on the caller's stream), and merges the `world` slots.  A rank may hold no rows at all (n_mem = 0).
 * Only for memory-update calls of ONE view on ONE scene in MUST3R_MEM_KV mode against a non-empty (global) memory: the per-frame call of the streaming schedule
 * (engine/inference.py:232-366).  The new K|V rows are appended to THIS rank's buffers as usual; the caller decides which rank keeps them
 * (must3r_amd.parallel.run_video_sharded(context_parallel=True): the frame's owner keeps, the others rewind).
 * `exchange` runs on the calling thread between launches on `stream`: it must enqueue the collective in stream order and must not synchronise the device with the
 * library's launches still queued behind it unless it has to (a host-staged exchange may).  Non-zero return aborts the call (status 1). */
typedef int (*must3r_hip_cp_exchange_fn)(void* user, int layer, void* slots, size_t slot_bytes, int n_slots, int my_slot, void* stream);
typedef struct must3r_hip_cp {
    int32_t world, rank;     /* ranks the memory is sharded over (>= 1; 1 = a group of one rank, the exchange still runs), this rank */
    int32_t n_mem_total;     /* memory rows over ALL ranks before this call (> 0); must3r_hip_decode_args.n_mem = THIS rank's rows (>= 0) */
    int32_t partial16;       /* 0: fp32 partials (un-normalised O); 1: the 16-bit partial format of the library's own split-KV path (O / l in the operand type + fp32
                              * (m, l)): half the bytes on the links, one more 16-bit rounding of an intermediate (inside the precision mode's tolerance; tests) */
    void* slots;             /* device buffer of world x slot_bytes bytes, 16-byte aligned; slot r = rank r's partial of the layer being exchanged */
    size_t slot_bytes;       /* >= must3r_hip_cp_slot_bytes(ctx, rows of the call) (partial16: must3r_hip_cp_slot_bytes16), a multiple of 16 */
    must3r_hip_cp_exchange_fn exchange;
    void* user;
} must3r_hip_cp;

typedef struct must3r_hip_decode_args {
    int32_t dtype;        /* MUST3R_BF16 / MUST3R_F16 / MUST3R_F16_W2 / MUST3R_F16_WA (the default of the Python modules: fp16 operands, split weights
                           * in the attention-side Linears, plain in the Mlp Linears): operand type AND element type of the memory buffers (fp16 for the
                           * three F16 modes);
                           * | MUST3R_ATTN_FP8: e4m3 Q / K, MUST3R_MEM_KV memory rows are [K e4m3 | V 16-bit] = 3*dec_dim bytes */
    int32_t mem_mode;     /* MUST3R_MEM_* */
    int32_t render;       /* decoder.py:267 `render`: memory is read-only, no exclusion mask */
    int32_t first_call;   /* current_mem is None: view 0 of group 0 gets no image2_embed (decoder.py:280-282) */
    int32_t n_groups;
    const must3r_hip_group* groups;
    int32_t n_mem;        /* Nm: valid memory tokens (per scene) before this call */
    /* per decoder layer: 16-bit ([K e4m3 | V 16-bit] byte rows with MUST3R_ATTN_FP8 + MUST3R_MEM_KV) [mem_capacity, mem_dim] row-major, mem_dim = 2*dec_dim (KV)
     * or dec_dim.  Rows [0,n_mem) are read; unless render, rows [n_mem, n_mem + sum(n_views*n_tokens)) are WRITTEN -- the in-place
     * form of torch.concatenate at decoder.py:239/330.  The call is refused when they would not fit mem_capacity. */
    void* const* mem;
    /* optional (`return_feats=True`, decoder.py:344-347 / :258-262): fp32 [dec_depth][R][dec_dim], R = n_scenes * sum(n_views*n_tokens),
     * rows scene-major, then group order; entry l = the residual stream after decoder block l, the last one after norm_dec
     * (decoder.py:150).  NULL = not wanted.  (feats[0] of the reference, the encoder tokens, is the caller's own input.) */
    float* feats;
    /* ---- ABI 5 ---- */
    int32_t mem_capacity; /* rows every scene's buffer can hold (> 0): bounds-checked against n_mem + the rows this call appends */
    /* B of the reference's tensors (decoder.py:170: x[i] is [B, nimg, Ni, Denc]): n_scenes independent scenes of identical shapes
     * (same groups, same n_mem) decoded by ONE launch sequence -- M = B x rows in every GEMM, B x views in the attention tables.
     * Scene b's memory of layer l is mem[l] + b * mem_scene_stride rows (mem_scene_stride >= mem_capacity); the scenes never
     * interact, results are bit-identical to B calls with n_scenes = 1 wherever the kernels' tile shapes coincide and within the
     * mode's tolerance otherwise.  0 / 1 = one scene. */
    int32_t n_scenes;
    int64_t mem_scene_stride;
    /* ---- ABI 8 ---- */
    const must3r_hip_cp* cp;   /* NULL: off.  Context-parallel cross attention (above): `mem` / `n_mem` describe this rank's SHARD of the memory */
    /* CausalMUSt3R.forward (decoder.py:435-553; the class the checkpoints are trained as), memory dropout off: in a memory update of several views, view i
     * cross-attends the old memory and the new (pre-feedback) tokens of the views BEFORE it in the call -- make_attn_mask decoder.py:389-433 -- instead of the new
     * tokens of every other view; when the memory is empty view 0 attends view 1's tokens (decoder.py:399-402).  One group (the class has no list dispatch); render
     * and one-view calls are MUSt3R's.  The tuple's tail (protected images / tokens, decoder.py:461-464) is the caller's bookkeeping. */
    int32_t causal;
} must3r_hip_decode_args;

/* bytes of one rank's partial for a context-parallel call of `rows` token rows: rows x (dec_dim + 2 x dec_heads) floats, rounded up to 256;
 * ...16: with partial16 = 1: rows x (2 dec_dim bytes + 2 x dec_heads floats) */
size_t must3r_hip_cp_slot_bytes(const must3r_hip_ctx* ctx, int rows);
size_t must3r_hip_cp_slot_bytes16(const must3r_hip_ctx* ctx, int rows);

/* MUSt3R.forward / forward_list (decoder.py:158-350).  Render calls whose view tables exceed the library's staging slot
 * (1365 views) are cut into ranges of scenes / views inside the library: rendered views are independent. */
int must3r_hip_decode(must3r_hip_ctx* ctx, const must3r_hip_decode_args* args, void* stream);

/* postprocess activation (engine/inference.py:19-27; tools/geometry.py:14-18): pointmaps fp32 [npix,7]
 *   -> pts3d [npix,3], pts3d_local [npix,3], conf [npix] */
int must3r_hip_postprocess(const float* pointmaps, float* pts3d, float* pts3d_local, float* conf, size_t npix,
                           void* stream);
/* the same with the activation named (ABI 6): ActivationType of must3r/model/blocks/head.py:8-21 -- NORM_EXP as above, LINEAR leaves
 * channels 0:3 / 3:6 as they are; conf = 1 + exp(ch 6) in both (engine/inference.py:26-27). */
enum { MUST3R_ACT_NORM_EXP = 0, MUST3R_ACT_LINEAR = 1 };
int must3r_hip_postprocess_act(const float* pointmaps, int activation, float* pts3d, float* pts3d_local, float* conf, size_t npix,
                               void* stream);

/* postprocess(..., compute_cam=True) (engine/inference.py:16-48), SURVEY.md section 8f rank 1: the activation above
 * plus, per view, focal = dust3r estimate_focal_knowing_depth(pts3d_local, pp=(W/2,H/2), 'weiszfeld')
 * (engine/inference.py:33-35) and (R, T) = roma.rigid_points_registration(pts3d_local -> pts3d, weights conf-1)
 * (engine/inference.py:37-40) written as c2w [n_views,4,4] row-major (engine/inference.py:42-46).
 * pointmaps fp32 [n_views,H,W,7]; outputs fp32; scratch: >= must3r_hip_postprocess_cam_scratch_bytes device bytes.
 * Uses a cooperative launch (all blocks co-resident). */
size_t must3r_hip_postprocess_cam_scratch_bytes(int n_views, int H, int W);
int must3r_hip_postprocess_cam(const float* pointmaps, int n_views, int H, int W, float* pts3d, float* pts3d_local,
                               float* conf, float* focal, float* c2w, void* scratch, size_t scratch_bytes, void* stream);
int must3r_hip_postprocess_cam_act(const float* pointmaps, int activation, int n_views, int H, int W, float* pts3d, float* pts3d_local,
                                   float* conf, float* focal, float* c2w, void* scratch, size_t scratch_bytes, void* stream);

/* Retrieval front-end on the encoder tokens, SURVEY.md section 8f rank 4 (retrieval/model.py).
 * must3r_hip_affine: out[M,N] fp32 = (A[M,K] - sub[K]) . B + bias[N] + resid[M,N]; with is_double the subtraction, the
 *   products and the sums are float64 like Whitener.forward (retrieval/model.py:67-79: x.double() - m, matmul with p), with
 *   b_transposed B is an nn.Linear weight [N,K] (the projector, :139-151,169).  sub / bias / resid may be NULL; sub, B and
 *   bias are float64 arrays when is_double, fp32 otherwise.
 * must3r_hip_row_norm: attention = x.norm(dim=-1) (:130-131).
 * must3r_hip_l2_normalize: F.normalize(x, dim) of a contiguous fp32 tensor seen as [outer, L, inner] (eps 1e-12):
 *   Whitener(l2norm=dim) (:77-78).  out may alias x.
 * must3r_hip_layernorm_act_f32: nn.LayerNorm(C, eps) (+ nn.GELU, erf form, when gelu) on fp32 rows: the hidden layers of a
 *   multi-layer projector (build_projector :139-151, Linear - LayerNorm - GELU stacks); gamma / beta may be NULL.
 * must3r_hip_topk_gather: how_select_local (:91-101): per image the k tokens of largest attention, sorted descending
 *   (ties: lower index first), their features, attentions and int64 indices.  N <= 4096.
 * must3r_hip_weighted_spoc: weighted_spoc (:82-88): normalize(sum_n attn[n] * feat[n,:]). */
int must3r_hip_affine(int is_double, const float* A, const void* sub, const void* B, int b_transposed, const void* bias,
                      const float* resid, float* out, int M, int N, int K, void* stream);
int must3r_hip_row_norm(const float* x, int M, int C, float* out, void* stream);
int must3r_hip_l2_normalize(const float* x, int64_t outer, int L, int64_t inner, float* out, void* stream);
int must3r_hip_layernorm_act_f32(const float* x, const float* gamma, const float* beta, float eps, int M, int C, int gelu,
                                 float* out, void* stream);
int must3r_hip_topk_gather(const float* feat, const float* attn, int n_images, int N, int C, int k, float* out_feat,
                           float* out_attn, int64_t* out_idx, void* stream);
int must3r_hip_weighted_spoc(const float* feat, const float* attn, int n_images, int N, int C, float* out, void* stream);

/* SLAM keyframe test, SURVEY.md section 8f rank 3 (slam/model.py:62-91 get_overlap_score; slam/nns.py:40-92).
 * must3r_hip_nn_query replaces KDTree_scipy.query (nns.py:52-57: scipy KDTree.query(k=1), Euclidean): out_dist[i] =
 * min_j |q_i - db_j| for fp32 xyz points [n,3] on the device, +inf when n_db == 0 (nns.py:53-54).  Exact (brute force).
 * must3r_hip_quadrant_ids replaces get_quadrant_id (slam/tools.py:9-31) on rays p - cam_center (nns.py:81,88):
 * out[i] in [0, 2*divider^2). */
int must3r_hip_nn_query(const float* db_xyz, int64_t n_db, const float* q_xyz, int64_t n_q, float* out_dist, void* stream);
int must3r_hip_quadrant_ids(const float* pts_xyz, int64_t n, const float* cam_center_host3, int divider, int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Operator-level entry points (the same kernels the two forwards are built from), exported so that parity
 * tests and roofline measurements can drive each kernel alone.
 * ------------------------------------------------------------------------------------------------------ */
enum { MUST3R_EPI_STORE16 = 0, MUST3R_EPI_STORE16_GELU = 1, MUST3R_EPI_QKV_ROPE = 2, MUST3R_EPI_RESID_F32 = 3,
       MUST3R_EPI_F32 = 4, MUST3R_EPI_HEAD = 5 };

/* out[M,N] = epi(A[M,K] . W[N,K]^T + bias): nn.Linear (+ fused epilogue).  A, W 16-bit.  N, K multiples of 64; lda % 8 == 0; ldc % 4 == 0
 * (16-bit outputs: ldc % 8 == 0 and a 16-byte aligned `out` -- the epilogue stores 16 bytes per lane). */
int must3r_hip_op_gemm(int dtype, int epi, const void* A, const void* W, const float* bias, void* out,
                       int M, int N, int K, int lda, int ldc,
                       const int64_t* pos, const float* rope_tab, int rope_cols, int rope_npos, /* QKV_ROPE */
                       const float* bias2, int row_start2, int accumulate,                      /* F32 / HEAD */
                       int ntok, int gw, int H, int W_img,                                      /* HEAD */
                       int wsplit, /* 2: W is [N, 2K] = [W_hi | W_lo], out = A W_hi^T + A W_lo^T; else 0 */
                       void* stream);
/* ABI 7 (r05).  Split weights whose LOW part is 2:4-sparse: in every group of 4 consecutive k of a weight row the 2 entries of largest magnitude of
 * W_lo = fp16(W - fp16(W)) are kept, so that the chip-filling kernel multiplies the low part of a 64-deep K-tile with ONE sparse MFMA per output fragment
 * (v_smfmac_f32_16x16x64_f16, twice the dense rate).  The library packs and uses this copy by itself for the split weights of a
 * context (M3R_SPARSE_LO=0: never); these two entry points drive the pair alone (tests, probes).
 *   pack : w fp32 [rows, K] (rows % 32 == 0, K % 64 == 0) -> vals fp16 [K/64][rows][32], idx uint32 [K/64][rows/32][64]
 *   gemm : out = epi(A . (W_hi + sparse W_lo)^T + bias); W2 = the dense [N, 2K] = [W_hi | W_lo] rows (hi half read), fp16 operands; N % 128 == 0, K % 64 == 0;
 *          launches too small to fill the chip with 256 x 128 tiles run the dense two-pass kernels on W2 instead. */
int must3r_hip_op_sparse24_pack(const float* w, int rows, int K, void* vals, void* idx, void* stream);
int must3r_hip_op_gemm_sp(int epi, const void* A, const void* W2, const void* Wlo_sp, const void* Widx_sp, const float* bias, void* out,
                          int M, int N, int K, int lda, int ldc, const int64_t* pos, const float* rope_tab, int rope_cols, int rope_npos, void* stream);
/* "LN fold" (one-view memory update): the LayerNorm between two Linears is applied AFTER the second product instead of before it.
 * Producer role (epi = RESID_F32 / F32): as must3r_hip_op_gemm, and additionally the new fp32 rows rounded to fp16 (x16_out, row stride
 * ldc), an optional fp32 copy (copy32_out) and per row and 16-column fragment (sum x, sum x^2) into stats_out [M][N/16][2].
 * Consumer role (epi = STORE16 / STORE16_GELU / QKV_ROPE; ln_stats != NULL): A = those raw fp16 rows, W2 = split fp16 copy of gamma (.) W,
 * bias = W beta + b, ln_s[n] = sum_k (gamma (.) W)[n][k]; out = epi(rstd_m (A W2^T - mu_m ln_s) + bias) = epi(LN(x) W^T + b).
 * ln_shift [M] (optional): the producer rounds and sums x - ln_shift[m] (an estimate of the row mean: the mean the previous consumer
 * measured), the consumer adds the mean it measures to it (ln_shift_init: the buffer holds nothing yet).
 * must3r/model/blocks/layers.py:91-99 (norm1 -> attn.qkv, norm2 -> cross_attn.projq, norm3 -> mlp.fc1). */
int must3r_hip_op_gemm_lnfold(int dtype, int epi, const void* A, const void* W2, const float* bias, void* out, int M, int N, int K,
                              int lda, int ldc, void* x16_out, float* copy32_out, float* stats_out, const float* ln_stats,
                              const float* ln_s, float ln_eps, float* ln_shift, int ln_shift_init, const int64_t* pos,
                              const float* rope_tab, int rope_cols, int rope_npos, float out_scale, int scale_cols, void* stream);
/* ABI 8 (r06).  The same fold on the chip-filling 256 x 256 tiles (batched decoder calls, encoder chunks; fp16; the library uses it by itself in MUST3R_F16_WA mode,
 * M3R_LNFOLD256=0: never).  wsplit = 2: W = [N, 2K] split rows + the packed sparse low part (must3r_hip_op_sparse24_pack), epi STORE16 / QKV_ROPE (consumer) or
 * RESID_F32 (producer); wsplit = 0: plain fp16 W [N, K], epi STORE16_GELU (consumer) or RESID_F32 (producer).  N % 256 == 0.
 * Producer: M % 256 == 0; x16_out rows (stride ldc) of x - ln_shift[m], (sum, sum of squares) per row and 64-COLUMN wave tile into stats_out [M][N/64][2], optional
 * copy32_out.  Consumer: ln_stats in that layout ([M][K/64][2]; K = 768 or 1024); ln_shift [M] (required; optional on producers) must hold valid values: += the measured mean. */
int must3r_hip_op_gemm_fold256(int epi, int wsplit, const void* A, const void* W, const void* Wlo_sp, const void* Widx_sp, const float* bias, void* out,
                               int M, int N, int K, int lda, int ldc, void* x16_out, float* copy32_out, float* stats_out, const float* ln_stats,
                               const float* ln_s, float ln_eps, float* ln_shift, const int64_t* pos, const float* rope_tab, int rope_cols, int rope_npos,
                               float out_scale, int scale_cols, void* stream);
/* cos/sin table fp32 [npos][16][2] for RoPE2D(freq, F0) with head dim 64 (host pointer) */
int must3r_hip_rope_table(float freq, float f0, int npos, float* out_host);

/* softmax(Q K^T / 8) V per head of 64; views: int32 [n_views][6] = q_row0, nq, kv_row0, nk, skip_lo, skip_hi
 * (DEVICE pointer).  Q/K/V/O 16-bit with row strides in elements (dtype | MUST3R_ATTN_FP8: Q/K e4m3 bytes with ldq/ldk in bytes, V/O 16-bit).
 * A view's K / V rows must span less than 2 GiB (nk * ld * element size): the staging uses 32-bit byte offsets.
 * nsplit > 1 selects split-KV (flash-decoding) with `scratch` of must3r_hip_attention_scratch_bytes() bytes and
 * total_q_rows = max(q_row0 + nq); nsplit <= 1 needs neither. */
size_t must3r_hip_attention_scratch_bytes(int nsplit, int total_q_rows, int heads);
int must3r_hip_op_attention(int dtype, const void* Q, const void* K, const void* V, void* O,
                            int ldq, int ldk, int ldv, int ldo, int heads,
                            const int32_t* views_dev, int n_views, int max_nq,
                            int nsplit, void* scratch, int total_q_rows, void* stream);

/* y = LN(x (+ add)) * w + b over rows of C; optional outputs may be NULL */
int must3r_hip_op_layernorm(int dtype, const float* x, const float* add, const float* w, const float* b,
                            void* out16, void* out16_lo, float* out32, float* copy32, int M, int C, float eps,
                            void* stream);
int must3r_hip_op_im2col(int dtype, const float* img, void* out16, int n_views, int H, int W, void* stream);
int must3r_hip_op_cast(int dtype, const float* in, void* out16, void* out16_lo, size_t n, void* stream);

/* debug: lane -> element mapping of the gfx950 transposing LDS read the attention kernel relies on; writes 256 int16 */
int must3r_hip_debug_tr_probe(void* out256_i16_dev, void* stream);

/* timing hooks used by bench.py: per-stage HIP-event timers recorded on the call's stream */
int must3r_hip_set_profiling(must3r_hip_ctx* ctx, int enabled);
/* returns the number of records written (<= max); each record: name (<=31 chars), milliseconds, flops.
 * First the kernel CLASSES (gemm128, gemm64, attn_self, attn_cross, attn_combine, layernorm, misc), then (ABI 6) one row per kernel
 * symbol, names prefixed "k:" -- GEMMs "k:<family>/e<epilogue>/w<1 plain|2 split>/n<tile width>", attention "k:attn3/q32/cross" ... --
 * so that every row can be matched with one symbol of a rocprofv3 --kernel-trace of the same command. */
typedef struct must3r_hip_prof_record { char name[32]; double ms; double flops; int64_t calls; } must3r_hip_prof_record;
int must3r_hip_get_profile(must3r_hip_ctx* ctx, must3r_hip_prof_record* out, int max, int reset);

#ifdef __cplusplus
}
#endif
#endif /* MUST3R_HIP_H */
