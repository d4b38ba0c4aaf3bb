// Host-side launcher declarations for the hand-written gfx950 kernels (one process per GPU, explicit stream).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace m3r {

enum DType { DT_BF16 = 0, DT_F16 = 1 };

// ---------------------------------------------------------------------------------------------
// GEMM   out[M,N] = epilogue( A[M,K] (16-bit, row-major, lda) x W[N,K]^T (16-bit, row-major) + bias[N] )
// ---------------------------------------------------------------------------------------------
enum Epi {
    EPI_STORE16 = 0,   // out 16-bit [M,ldc]
    EPI_STORE16_GELU,  // out 16-bit, exact-erf GELU
    EPI_QKV_ROPE,      // out 16-bit, 2-D RoPE on columns < rope_cols (q and k of a fused qkv projection)
    EPI_RESID_F32,     // out fp32 [M,ldc] += acc + bias   (in-place residual stream update)
    EPI_F32,           // out fp32 = acc + bias (+ bias2 on rows >= row_start2) ; accumulate -> out += acc
    EPI_HEAD,          // fp32 pixel-shuffled scatter into [view][H][W][7] (weights pre-permuted) ; accumulate alike
    EPI_COUNT
};

// r06: the e4m3 attention path of BASELINE.json configs[4] (attn4_kernel<.., F8>, the [K e4m3 | V 16-bit] memory rows, the quantisation kernel) is PARKED: measured
// +0.4 % on the mixed-resolution scene at 1.2e-3 ... 1.4e-3 from the 16-bit path, i.e. outside the 1e-3 target with nothing to show for it.  It is
// compiled only with `make EXTRA=-DM3R_ATTN_FP8`; the default library refuses MUST3R_ATTN_FP8 with an error that says so (must3r_hip_has_fp8_attention() = 0).
#ifdef M3R_ATTN_FP8
constexpr bool kAttnFp8Built = true;
#else
constexpr bool kAttnFp8Built = false;
#endif

struct GemmArgs {
    const void* A;
    const void* W;
    const float* bias;
    void* out;
    int M, N, K, lda, ldc;
    // split-weight mode: W is [N, 2K] = [W_hi | W_lo] (both fp16); every K-tile multiplies the activation tile with
    // both halves: out = A.W_hi^T + A.W_lo^T in fp32 -> weight rounding error drops from 2^-11 to ~2^-22.
    int wsplit;              // 0/1 = plain, 2 = [hi|lo]
    // r05: a 2:4-sparse copy of the low part (launch_sparse24_pack) for the chip-filling split-weight kernel: in every group of 4 consecutive k of a row
    // the 2 entries of largest magnitude.  Wlo_sp [K/64][wsp_rows][32] 16-bit (K-tile-major: a tile's rows of one K-tile are contiguous),
    // Widx_sp [K/64][wsp_rows/32][64 lanes] dwords of 2-bit positions (lane = row % 16 + 16 * (k / 16 % 4); low half rows 0-15 of the 32, high half rows 16-31).
    // nullptr = not available (the launcher then runs the dense two-pass kernels).  wsp_rows = rows of the whole parameter (grouped launches index it).
    const void* Wlo_sp;
    const void* Widx_sp;
    int wsp_rows;
    // 16-bit-store epilogues: columns < scale_cols are multiplied by out_scale before rounding (the softmax scale
    // 1/sqrt(64) * log2(e) is folded into q here so the attention kernel works in the exp2 domain for free)
    float out_scale;         // 0 -> no scaling
    int scale_cols;
    // grouped launch: blockIdx.y = g selects A + g*strideA, W + g*strideW, bias + g*strideB and out_table[g]
    // (independent problems of equal shape, e.g. the per-layer K|V projections of one memory update)
    int batch;               // 0/1 = single problem
    long long strideA, strideW, strideB;   // in elements
    void* const* out_table;  // device array of `batch` output base pointers
    // wdiv > 1: problems g and g' with g / wdiv == g' / wdiv share weights and bias (W + (g / wdiv) * strideW): the S scenes of a
    // batched decoder call write their K|V rows of layer l into S different memory buffers with layer l's one projection
    int wdiv;
    // EPI_QKV_ROPE
    const int64_t* pos;      // [M,2] (y,x)
    const float* rope_tab;   // [npos][16][2] (cos,sin)
    int rope_cols;
    int rope_npos;
    // EPI_F32
    const float* bias2;
    int row_start2;
    int row_period2;         // > 0: bias2 on rows with (m % row_period2) >= row_start2 (batched scenes: the reference view of EVERY scene)
    int accumulate;          // EPI_F32 / EPI_HEAD: add into out, skip bias
    // EPI_HEAD: rows are `ntok`-token views of one aspect ratio
    int ntok, gw, H, Wimg;
    // r05: the views of the launch belong to scenes of `head_views` views each whose pointmaps are `head_scene_skip` elements further apart than contiguous
    // (must3r_hip_group::pointmaps_scene_stride); 0 / 0 = one contiguous [views, H, W, 7] block
    int head_views;
    long long head_scene_skip;
    // LayerNorm folded into the GEMMs around it (one-view memory update; DESIGN.md section 3, "LN fold"):
    //   producer (EPI_RESID_F32 / EPI_F32): besides `out` it writes the new fp32 rows rounded to the 16-bit type (x16_out, row stride ldc),
    //     an optional second fp32 copy (copy32_out) and, per row and 16-column fragment, (sum x, sum x^2) into stats_out [M][N/16][2];
    //   consumer (STORE16 / STORE16_GELU / QKV_ROPE): A = those raw 16-bit rows, W = gamma (.) W, `bias` = c = W beta + b, and
    //     out = epi( rstd_m (acc - mu_m s_n) + c_n ), (mu_m, rstd_m) from ln_stats [M][K/16][2], s_n = sum_k W'[n][k] (ln_s) -- which is
    //     LN(x) W^T + b with the normalisation applied after the product instead of before it.
    void* x16_out;
    float* copy32_out;
    float* stats_out;
    const float* ln_stats;
    const float* ln_s;
    float ln_eps;
    //   ln_shift [M]: a per-row estimate of the row mean (the mean the PREVIOUS consumer measured).  The producer rounds and sums
    //     y = x - shift instead of x, so a common offset of a row does not eat the fp16 mantissa of its small deviations and the one-pass
    //     variance works on shifted data; nothing changes for the consumer ((y - mean(y)) rstd = (x - mean(x)) rstd), except that its
    //     column-0 blocks leave shift += mean(y) (= the current row mean) for the next producer.  ln_shift_init: the buffer holds nothing yet.
    float* ln_shift;
    int ln_shift_init;
    // r06: the same fold on the chip-filling 256 x 256 tiles (gemm256s_kernel / gemm256p_kernel<.., WS = 1, BN = 256, FOLD = 1>; batched decoder calls and the encoder).
    //   fold256 = 1 on a producer (EPI_RESID_F32; needs M % 256 == 0, N % 256 == 0): x16_out rows + ONE (sum, sum of squares) per row and 64-column wave tile in
    //     stats_out [M][N/64][2] (+ copy32_out);  on a consumer (STORE16 / STORE16_GELU / QKV_ROPE; K = 768 or 1024): ln_stats is that [M][K/64][2] layout.
    //   ln_shift (required on consumers) must hold valid values (zeros at the start of a call): every consumer adds the mean it measured (ln_shift_init is ignored).
    int fold256;
    int gm;                  // set by the 256-row launchers from the option G256_GM: row-blocks per group of the tile walk (0 = 4)
#ifdef GEMM_TRACE
    int trace_block;         // probe builds only (scripts/probes/gemm256p_trace.hip): the block whose waves leave cycle stamps
#endif
};

// returns 0 on success, non-zero (and sets *err) on an unsupported shape
int launch_gemm(DType dt, Epi epi, const GemmArgs& a, hipStream_t s, const char** err);
// the kernel the calling thread's last launch_gemm picked: "<family>/e<EPI>/w<WS>/n<BN>" (one symbol of a kernel trace)
const char* gemm_last_kernel();
// r06: would launch_gemm route a launch of this shape to the 256 x 256 kernels that carry the LN fold (GemmArgs::fold256)?  split = [hi | lo] weights with a packed
// 2:4-sparse low part (gemm256s_kernel), else plain fp16 weights (gemm256p_kernel).  The model folds a call's LayerNorms only when every Linear around them says yes.
bool gemm_fold256_shape_ok(int M, int N, int K, bool split);

// ---------------------------------------------------------------------------------------------
// fused softmax attention, head dim 64, flash-style (no N x M score matrix in HBM)
// ---------------------------------------------------------------------------------------------
struct AttnView {
    int q_row0;   // first query row of this view in Q / O
    int nq;       // number of query tokens
    int kv_row0;  // first key row in K / V
    int nk;       // number of keys (before exclusion)
    int skip_lo;  // keys [skip_lo, skip_hi) (relative to kv_row0) are excluded (own-token rule,
    int skip_hi;  //   MUSt3R.make_mem_mask decoder.py:119-139); skip_lo == skip_hi -> none
};

struct AttnArgs {
    const void* Q; const void* K; const void* V; void* O;   // 16-bit
    int ldq, ldk, ldv, ldo;                                   // row strides in elements
    int heads;
    const AttnView* views;                                    // device pointer
    int nviews;
    // one-view launches (every attention of a one-view memory update): the view travels in the kernel arguments, which saves each block
    // the dependent global load of its table row in front of the Q / K / V address arithmetic (~1 us of prologue latency per launch)
    int view0_inline;                                         // 1: use view0 instead of views[0]
    AttnView view0;
    int max_nq;                                               // max over views of nq
    float scale;                                              // 1/sqrt(64); ignored when q_prescaled
    int q_prescaled;                                          // Q already carries scale*log2(e)
    // split-KV (flash-decoding): when a launch has too few (view, head, q-block) groups to fill 256 CUs the key range
    // is cut into nsplit chunks, each block writes un-normalised fp32 partials + (m, l) and a second kernel merges.
    int nsplit;            // <= 1: single pass
    float* part_o;         // [nsplit][total_q_rows][heads*64] fp32
    float* part_ml;        // [nsplit][total_q_rows][heads][2] fp32 (running max in log2 domain, row sum)
    int total_q_rows;      // max over views of q_row0 + nq
    int dense_rows;        // every row < total_q_rows belongs to a view of this launch (no (m,l) pre-fill needed)
    int part16;            // set by the launcher: part_o holds O_s / l_s in the 16-bit operand type instead of fp32 O_s
    int fp8;               // Q and K are OCP e4m3 bytes (ldq / ldk in BYTES): Q K^T runs on v_mfma_scale_f32_32x32x64_f8f6f4; V (ldv in elements), the
                           // softmax numerators and O stay 16-bit (attn4_kernel<.., F8 = true>; DESIGN.md section 4 for why V does)
    int max_nk;            // max over views of nk when the host knows it (0: unknown): lets the launcher refuse K / V spans of 2 GiB or more
    // r06 (context-parallel cross attention): distance between the partials of two splits / slots, in elements of part_o (fp32, or the 16-bit type with part16)
    // and in floats of part_ml; 0 = the dense default total_q_rows * heads * 64 / total_q_rows * heads * 2.  Read by the combine kernels only.
    long long part_stride_o, part_stride_ml;
};
// r06, context-parallel cross attention over a memory sharded across ranks (SURVEY.md section 8f "later"; decoder.py:301-321 with the keys of one view spread over
// `world` processes).  A rank's contribution to one layer is ONE partial per query row -- the split-KV partial format one level up:
//   p16 = 0: un-normalised O = sum_k 2^(s_k - m) v_k over ITS keys, fp32 [rows][heads*64];   p16 = 1: O / l in the 16-bit type [rows][heads*64] (half the link bytes);
//   then (m, l) = (reference in the log2 domain, row sum) fp32 [rows][heads][2].
//   launch_attention_partial_merge: the nsplit local split-KV partials of `a` (layout of launch_attention_phase(.., 1, ..)) -> one such partial in (out_o, out_ml)
//   launch_attention_partial_empty: the partial of a rank that holds no key: O = 0, (m, l) = (-inf, 0)
//   launch_attention_partial_final: `nslots` partials at slots_o + s * stride_o (elements of the partial's O type) / slots_ml + s * stride_ml (floats) -> a.O (ldo)
// One fp32 slot in, `final` reproduces bit for bit what launch_attention_phase(.., 2, ..) writes for the same splits (same sums in the same order, 2^0 = 1).
int launch_attention_partial_merge(DType dt, const AttnArgs& a, float* out_o, float* out_ml, int p16, hipStream_t s, const char** err);
int launch_attention_partial_empty(float* out_o, float* out_ml, int rows, int heads, int p16, hipStream_t s, const char** err);
int launch_attention_partial_final(DType dt, const AttnArgs& a, const float* slots_o, const float* slots_ml, long long stride_o, long long stride_ml, int nslots,
                                   int p16, hipStream_t s, const char** err);
// bytes of scratch launch_attention needs for a given split factor
size_t attention_split_scratch_bytes(int nsplit, int total_q_rows, int heads);
// heuristic split factor for a launch
int attention_pick_split(int nviews, int heads, int max_nq, int max_nk);
int launch_attention(DType dt, const AttnArgs& a, hipStream_t s, const char** err);
// launches too small to fill the chip with 128-query blocks run the 16-row-per-wave 16 x 16 kernel (16-bit operands only): the model
// does not quantise Q / K for those
bool attention_is_small(int nviews, int heads, int max_nq, int nsplit);
// the three launches of a split-KV attention, separately (profiling): (m,l) pre-fill, main kernel, combine
int launch_attention_phase(DType dt, const AttnArgs& a, int phase, hipStream_t s, const char** err);
const char* attention_last_kernel();   // main kernel of the calling thread's last phase-1 launch

// ---------------------------------------------------------------------------------------------
// row / elementwise kernels
// ---------------------------------------------------------------------------------------------
struct LnArgs {
    const float* x;      // [M,C] fp32 input, or nullptr to read x16 instead
    const void* x16;     // optional 16-bit input [M,C] (memory_mode 'raw': LayerNorm of stored 16-bit tokens)
    void* raw16;         // optional 16-bit copy of x (+add) before normalisation (memory_mode 'raw' rows)
    const float* add;    // optional [M,C] added to x before the statistics (feedback offset)
    const float* w; const float* b;
    void* out16;         // optional 16-bit [M,C]
    void* out16_lo;      // optional 16-bit residual part: T(y - float(T(y)))  (split-precision head GEMM)
    void* out16_dup;     // optional second copy of out16 (the head GEMM reads [y_hi | y_lo | y_hi] as one K = 3C operand)
    int ld16;            // row stride (elements) of out16 / out16_lo / out16_dup; 0 = C
    float* out32;        // optional fp32 [M,C]
    float* copy32;       // optional raw copy of x (+add) (memorised layer input, decoder.py:304-305)
    int M, C;
    float eps;
    // grouped rows: row r belongs to group g = r / rows_per_group: affine parameters w + g*C, b + g*C; `add` (shape
    // [rows_per_group, C]) is applied to groups < add_groups only (feedback offset: all layers but the last)
    int rows_per_group;      // 0 = ungrouped
    int add_groups;
};
int launch_layernorm(DType dt, const LnArgs& a, hipStream_t s, const char** err);

// img fp32 [V,3,H,W] -> patches 16-bit [V*gh*gw, 3*16*16]; column = c*256 + i*16 + j
int launch_im2col(DType dt, const float* img, void* out16, int V, int H, int W, hipStream_t s, const char** err);
// fp32 -> 16-bit (and optional low part)
int launch_cast(DType dt, const float* in, void* out16, void* out16_lo, size_t n, hipStream_t s, const char** err);
// 16-bit [rows, ld_in] -> e4m3 bytes [rows, ld_out] (clamped to +-448); optional per-group output table (rows_per_group rows each).
// tail_cols > 0: input columns [cols, cols + tail_cols) are copied verbatim (16-bit) behind the e4m3 bytes: output row =
// [cols bytes e4m3 | tail_cols x 16-bit] -- the K (e4m3) | V (16-bit) memory rows of the fp8 attention mode.  ld_out in BYTES.
int launch_quant8(DType dt, const void* in16, int ld_in, void* out8, int ld_out, void* const* out_table, int rows_per_group,
                  size_t rows, int cols, int tail_cols, hipStream_t s, const char** err);
// pos int64 [V, gh*gw, 2] = (y, x) row-major grid
int launch_fill_pos(int64_t* pos, int V, int gh, int gw, hipStream_t s, const char** err);
// pointmaps [npix,7] fp32 -> pts3d [npix,3], pts3d_local [npix,3], conf [npix]; linear: ActivationType.LINEAR instead of NORM_EXP
int launch_postprocess(const float* pm, int linear, float* pts3d, float* pts3d_local, float* conf, size_t npix, hipStream_t s,
                       const char** err);
// retrieval front-end on the encoder tokens (retrieval/model.py:59-101,165-183); retrieval.hip
int launch_gemmx(int is_double, const float* A, const void* sub, const void* B, int b_transposed, const void* bias, const float* resid,
                 float* out, int M, int N, int K, hipStream_t s, const char** err);
int launch_row_norm(const float* x, int M, int C, float* out, hipStream_t s, const char** err);
int launch_l2_normalize(const float* x, long long outer, int L, long long inner, float* out, hipStream_t s, const char** err);
int launch_ln_act_f32(const float* x, const float* gamma, const float* beta, float eps, int M, int C, int gelu, float* out, hipStream_t s,
                      const char** err);
int launch_topk_gather(const float* feat, const float* attn, int Bn, int N, int C, int k, float* out_feat, float* out_attn,
                       long long* out_idx, hipStream_t s, const char** err);
int launch_weighted_spoc(const float* feat, const float* attn, int Bn, int N, int C, float* out, hipStream_t s, const char** err);
// SLAM keyframe test (slam/nns.py, slam/tools.py:9-31): exact 1-NN distances by brute force, view-direction quadrants; nn.hip
int launch_nn_query(const float* db, long long n_db, const float* q, long long n_q, float* out_dist, hipStream_t s, const char** err);
int launch_quadrant_ids(const float* pts, long long n, const float* cam_center_host, int div, int* out, hipStream_t s, const char** err);
// postprocess(compute_cam=True): activation + focal (Weiszfeld) + weighted rigid registration, cam.hip
size_t cam_scratch_bytes(int n_views, int H, int W);
int launch_postprocess_cam(const float* pm, int linear, int n_views, int H, int W, float* pts3d, float* pts3d_local, float* conf,
                           float* focal, float* c2w, void* scratch, size_t scratch_bytes, hipStream_t s, const char** err);
// 16-bit weight low part: lo = T(w - float(T(w)))  and hi = T(w), from fp32
int launch_split16(DType dt, const float* in, void* hi, void* lo, size_t n, hipStream_t s, const char** err);
// 2:4-sparse copy of the fp16 low part of w fp32 [rows, K] (rows % 32 == 0, K % 64 == 0): vals [K/64][rows][32] fp16, idx [K/64][rows/32][64] dwords (GemmArgs::Wlo_sp)
int launch_sparse24_pack(const float* w, int rows, int K, void* vals, void* idx, hipStream_t s, const char** err);

// debug: mapping of ds_read_b64_tr_b16 (out: 256 shorts)
int launch_tr_probe(short* out, hipStream_t s);

}  // namespace m3r
