// Native host runtime of libmust3r_hip: context, weight store, workspace arena and the two forwards
// (per-view ViT-L encoder, ViT-B memory decoder) expressed as launches of the gfx950 kernels in
// gemm.hip / attention.hip / misc.hip on the caller's HIP stream.  C ABI: include/must3r_hip.h.
//
// Algorithm citations are to the reference tree (must3r/model/...):
//   encoder.py:46-52, blocks/layers.py:51-54,90-99, blocks/attention.py:37-149, decoder.py:158-350,
//   feedback_mechanism.py:39-53, blocks/head.py:63-72, tools/image.py:9-14.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/must3r_hip.h"
#include "kernels.hpp"
#include "options.hpp"

using namespace m3r;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
#define M3R_OK(expr)                  \
    do {                              \
        int rc__ = (expr);            \
        if (rc__) return rc__;        \
    } while (0)

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct Param {
    std::vector<int64_t> shape;
    size_t n = 0;
    float* d = nullptr;       // fp32 master on device
    void* h16[2] = {nullptr, nullptr};   // packed 16-bit copy per dtype (lazy)
    void* l16[2] = {nullptr, nullptr};   // low part (split precision), lazy, only where needed
    void* x2[2] = {nullptr, nullptr};    // [rows][hi(K) | lo(K)] layout for the split-weight GEMM, lazy
    void* x3[2] = {nullptr, nullptr};    // [rows][hi | hi | lo] layout: one K = 3 in launch of the split-precision head, lazy
    void* sp_vals = nullptr;             // r05: 2:4-sparse copy of the fp16 low part (GemmArgs::Wlo_sp / Widx_sp), built with x2[DT_F16] where the shape allows
    void* sp_idx = nullptr;
    bool loaded = false;
    bool derived = false;     // built by finalize, not loaded
    bool optional = false;    // may legitimately be absent (feedback layer variants)
};

// softmax scale folded into q by the projection epilogues: 1/sqrt(64) * log2(e)
static const float kQScale = 0.125f * 1.44269504088896340736f;

struct ProfEntry { hipEvent_t a, b; int cat; double flops; char kern[40]; };
struct ProfKern { double ms = 0, flops = 0; long long calls = 0; };
enum ProfCat { PC_GEMM128 = 0, PC_GEMM64, PC_ATTN_SA, PC_ATTN_CA, PC_ATTN_COMBINE, PC_LN, PC_MISC, PC_COUNT };
static const char* kProfNames[PC_COUNT] = {"gemm128", "gemm64", "attn_self", "attn_cross", "attn_combine", "layernorm", "misc"};

// parameters of one transformer block, resolved to Param* once per context
enum LayerField { LF_N1W, LF_N1B, LF_QKVW, LF_QKVB, LF_PROJW, LF_PROJB, LF_N2W, LF_N2B, LF_FC1W, LF_FC1B, LF_FC2W, LF_FC2B, LF_NYW, LF_NYB, LF_PQW, LF_PQB, LF_PKVW, LF_PKVB, LF_CPW, LF_CPB, LF_N3W, LF_N3B, LF_QKVLN_W, LF_QKVLN_S, LF_QKVLN_C, LF_PQLN_W, LF_PQLN_S, LF_PQLN_C, LF_FC1LN_W, LF_FC1LN_S, LF_FC1LN_C, LF_COUNT };
static const char* kLayerFieldNames[LF_COUNT] = {"norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias", "norm_y.weight", "norm_y.bias", "cross_attn.projq.weight", "cross_attn.projq.bias", "cross_attn.projkv.weight", "cross_attn.projkv.bias", "cross_attn.proj.weight", "cross_attn.proj.bias", "norm3.weight", "norm3.bias", "attn.qkv_ln.weight", "attn.qkv_ln.s", "attn.qkv_ln.c", "cross_attn.projq_ln.weight", "cross_attn.projq_ln.s", "cross_attn.projq_ln.c", "mlp.fc1_ln.weight", "mlp.fc1_ln.s", "mlp.fc1_ln.c"};

struct must3r_hip_ctx {
    must3r_hip_config cfg;
    int device = 0;
    std::map<std::string, Param> params;
    bool fin_enc = false, fin_dec = false;
    // per-layer parameter tables (resolved once: no string building / map lookups per launch); entries of absent parameters are null
    std::vector<std::vector<struct Param*>> enc_tab, dec_tab;
    int wsplit = 0;            // 2 while a forward runs in MUST3R_F16_W2 / MUST3R_F16_WA mode
    int mlp_plain = 0;         // 1 in MUST3R_F16_WA mode: the Mlp Linears take plain fp16 weights (one MFMA pass)
    int attn8 = 0;             // 1 while a forward runs with MUST3R_ATTN_FP8 (e4m3 attention operands)
    float* rope_tab = nullptr;
    int rope_npos = 0;
    // workspace arena (grow-only)
    char* ws = nullptr;
    size_t ws_cap = 0, ws_off = 0;
    // pinned staging for per-call view tables
    static constexpr int kSlots = 32;
    static constexpr size_t kSlotBytes = 64 * 1024;
    char* pin = nullptr;
    char* pin_dev = nullptr;
    hipEvent_t slot_ev[kSlots];
    bool slot_used[kSlots];
    int slot_next = 0;
    // profiling
    bool prof = false;
    std::vector<ProfEntry> prof_entries;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[PC_COUNT] = {0}, prof_flops[PC_COUNT] = {0};
    long long prof_calls[PC_COUNT] = {0};
    std::map<std::string, ProfKern> prof_kern;   // per kernel symbol (GEMM: family / epilogue / weight layout / tile width; attention: kernel + self / cross)
    // r05: split-weight matrix ([hi | lo] rows, Param::x2) -> its 2:4-sparse low part; the gemm() wrapper hands it to the launcher, which uses it in the
    // chip-filling kernel only (every other split-weight kernel reads the dense low half of the rows)
    struct SparseLo { const void* vals; const void* idx; int rows; };
    std::map<const void*, SparseLo> sparse_lo;
};

// The entry points run on the context's device and leave the caller's current device as they found it (a process that
// drives several GPUs keeps its own current device; torch's notion of it is not changed behind its back).
struct DeviceGuard {
    int prev = -1, want;
    explicit DeviceGuard(int d) : want(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) (void)hipSetDevice(want);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
    }
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int ws_reserve(must3r_hip_ctx* c, size_t bytes, hipStream_t s) {
    if (bytes <= c->ws_cap) { c->ws_off = 0; return 0; }
    if (c->ws) {
        HIP_OK(hipStreamSynchronize(s));
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipFree(c->ws));
        c->ws = nullptr;
    }
    const size_t cap = align_up(bytes + bytes / 8, 1 << 20);
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&c->ws), cap));
    c->ws_cap = cap;
    c->ws_off = 0;
    return 0;
}
template <class T> static T* ws_take(must3r_hip_ctx* c, size_t count) {
    const size_t off = align_up(c->ws_off, 256);
    c->ws_off = off + count * sizeof(T);
    if (c->ws_off > c->ws_cap) return nullptr;  // cannot happen if reserve() was sized with ws_need()
    return reinterpret_cast<T*>(c->ws + off);
}
static size_t ws_need(size_t cur, size_t count, size_t elt) { return align_up(cur, 256) + count * elt; }

// upload a small host table to device memory that stays valid until the stream has consumed it
static int upload_table(must3r_hip_ctx* c, const void* host, size_t bytes, void** dev, hipStream_t s) {
    if (bytes > must3r_hip_ctx::kSlotBytes) return fail("view table too large (%zu bytes)", bytes);
    const int i = c->slot_next;
    c->slot_next = (i + 1) % must3r_hip_ctx::kSlots;
    if (c->slot_used[i]) HIP_OK(hipEventSynchronize(c->slot_ev[i]));
    char* h = c->pin + (size_t)i * must3r_hip_ctx::kSlotBytes;
    char* d = c->pin_dev + (size_t)i * must3r_hip_ctx::kSlotBytes;
    memcpy(h, host, bytes);
    // (r06, measured and dropped: a one-block copy KERNEL reading the pinned slot instead of this copy-engine transfer -- the 28 gaps of ~35 us per one-scene pass that
    // surround these uploads became ~40 gaps of 6-40 us, 59.3 vs 58.7 ms per pass: scripts/r06_runs/c08_tables.sh, profiles/r06_single_scene_seams.txt)
    HIP_OK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
    HIP_OK(hipEventRecord(c->slot_ev[i], s));
    c->slot_used[i] = true;
    *dev = d;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// profiling helpers
// ------------------------------------------------------------------------------------------------
static hipEvent_t ev_get(must3r_hip_ctx* c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope {
    must3r_hip_ctx* c; hipStream_t s; ProfEntry e; bool on;
    ProfScope(must3r_hip_ctx* c_, hipStream_t s_, int cat, double flops) : c(c_), s(s_), on(c_ && c_->prof) {
        if (on) { e.a = ev_get(c); e.b = ev_get(c); e.cat = cat; e.flops = flops; e.kern[0] = 0; (void)hipEventRecord(e.a, s); }
    }
    void kernel(const char* name, const char* suffix = "") { if (on) snprintf(e.kern, sizeof(e.kern), "%s%s", name, suffix); }
    ~ProfScope() { if (on) { (void)hipEventRecord(e.b, s); c->prof_entries.push_back(e); } }
};
static void prof_flush(must3r_hip_ctx* c) {
    for (auto& e : c->prof_entries) {
        (void)hipEventSynchronize(e.b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e.a, e.b);
        c->prof_ms[e.cat] += ms;
        c->prof_flops[e.cat] += e.flops;
        c->prof_calls[e.cat] += 1;
        if (e.kern[0]) {
            ProfKern& k = c->prof_kern[e.kern];
            k.ms += ms; k.flops += e.flops; k.calls += 1;
        }
        c->ev_pool.push_back(e.a);
        c->ev_pool.push_back(e.b);
    }
    c->prof_entries.clear();
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
// ws_override >= 0: the caller chose the weight layout of THIS launch (wmlp below); otherwise the context's mode decides
static int gemm(must3r_hip_ctx* c, DType dt, Epi epi, GemmArgs a, hipStream_t s, int ws_override = -1) {
    const char* err = "";
    const long nbat = a.batch > 1 ? a.batch : 1;
    const int ws = ws_override >= 0 ? ws_override : ((c && epi != EPI_HEAD) ? c->wsplit : a.wsplit);
    const long tiles128 = (long)((a.M + 127) / 128) * (a.N / 128) * nbat;
    const int cat = ws == 2 ? (((long)((a.M + 127) / 128) * (a.N / 64) * nbat >= 384) ? PC_GEMM128 : PC_GEMM64)
                            : ((a.N % 128 == 0 && tiles128 >= 192) ? PC_GEMM128 : PC_GEMM64);
    if (ws_override >= 0) a.wsplit = ws_override;
    else if (c && epi != EPI_HEAD) a.wsplit = c->wsplit;
    if (c && a.wsplit == 2 && c->mlp_plain && epi != EPI_HEAD && !c->sparse_lo.empty() && opt(OPT_SPARSE_LO) != 0) {   // (the sparse copies of an earlier MUST3R_F16_WA forward are not for MUST3R_F16_W2)
        auto it = c->sparse_lo.find(a.W);
        if (it != c->sparse_lo.end()) { a.Wlo_sp = it->second.vals; a.Widx_sp = it->second.idx; a.wsp_rows = it->second.rows; }
    }
    ProfScope ps(c, s, cat, 2.0 * a.M * a.N * a.K * (a.batch > 1 ? a.batch : 1));
    if (launch_gemm(dt, epi, a, s, &err)) return fail("%s (M=%d N=%d K=%d epi=%d)", err, a.M, a.N, a.K, (int)epi);
    ps.kernel(gemm_last_kernel());
    return 0;
}
static GemmArgs gargs(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int lda, int ldc) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.out = out; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc;
    return a;
}
static LnArgs lnargs(const float* x, const float* add, const float* w, const float* b, void* o16, void* o16lo, float* o32,
                     float* copy, int M, int C, float eps) {
    LnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.add = add; a.w = w; a.b = b; a.out16 = o16; a.out16_lo = o16lo; a.out32 = o32; a.copy32 = copy;
    a.M = M; a.C = C; a.eps = eps;
    return a;
}
static int layernorm_a(must3r_hip_ctx* c, DType dt, const LnArgs& a, hipStream_t s) {
    const char* err = "";
    ProfScope ps(c, s, PC_LN, 0.0);
    if (launch_layernorm(dt, a, s, &err)) return fail("%s", err);
    return 0;
}
static int layernorm(must3r_hip_ctx* c, DType dt, const float* x, const float* add, const float* w, const float* b,
                     void* o16, void* o16lo, float* o32, float* copy, int M, int C, float eps, hipStream_t s) {
    return layernorm_a(c, dt, lnargs(x, add, w, b, o16, o16lo, o32, copy, M, C, eps), s);
}
static int quant8(must3r_hip_ctx* c, DType dt, const void* in16, int ld_in, void* out8, int ld_out, void* const* out_table,
                  int rows_per_group, size_t rows, int cols, int tail_cols, hipStream_t s) {
#ifdef M3R_ATTN_FP8
    const char* err = "";
    ProfScope ps(c, s, PC_MISC, 0.0);
    if (launch_quant8(dt, in16, ld_in, out8, ld_out, out_table, rows_per_group, rows, cols, tail_cols, s, &err)) return fail("%s", err);
    return 0;
#else
    return fail("quant8: the e4m3 attention path is an experiment build (make EXTRA=-DM3R_ATTN_FP8)");
#endif
}
static int attention(must3r_hip_ctx* c, DType dt, const AttnArgs& a, double flops, int cat, hipStream_t s) {
    const char* err = "";
    if (a.nsplit > 1 && !a.dense_rows) {
        ProfScope ps(c, s, PC_ATTN_COMBINE, 0.0);
        if (launch_attention_phase(dt, a, 0, s, &err)) return fail("%s", err);
    }
    {
        ProfScope ps(c, s, cat, flops);   // the attention kernel alone: what rocprofv3 reports under its symbol
        if (launch_attention_phase(dt, a, 1, s, &err)) return fail("%s", err);
        ps.kernel(attention_last_kernel(), cat == PC_ATTN_SA ? "/self" : "/cross");
    }
    if (a.nsplit > 1) {
        ProfScope ps(c, s, PC_ATTN_COMBINE, 0.0);
        if (launch_attention_phase(dt, a, 2, s, &err)) return fail("%s", err);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// parameters
// ------------------------------------------------------------------------------------------------
static void expect(must3r_hip_ctx* c, const std::string& name, std::vector<int64_t> shape, bool optional = false) {
    Param p;
    p.optional = optional;
    p.shape = shape;
    p.n = 1;
    for (auto d : shape) p.n *= (size_t)d;
    c->params[name] = p;
}
static void expect_linear(must3r_hip_ctx* c, const std::string& pfx, int64_t out, int64_t in, bool optional = false) {
    expect(c, pfx + ".weight", {out, in}, optional);
    expect(c, pfx + ".bias", {out}, optional);
}
static void expect_ln(must3r_hip_ctx* c, const std::string& pfx, int64_t dim, bool optional = false) {
    expect(c, pfx + ".weight", {dim}, optional);
    expect(c, pfx + ".bias", {dim}, optional);
}
static bool is_loaded(must3r_hip_ctx* c, const std::string& name) {
    auto it = c->params.find(name);
    return it != c->params.end() && it->second.loaded;
}

static void build_expected(must3r_hip_ctx* c) {
    const must3r_hip_config& g = c->cfg;
    const int64_t C = g.enc_dim, D = g.dec_dim, p = g.patch_size, r = g.mlp_ratio;
    expect(c, "encoder.patch_embed.proj.weight", {C, 3, p, p});
    expect(c, "encoder.patch_embed.proj.bias", {C});
    for (int i = 0; i < g.enc_depth; ++i) {
        const std::string b = "encoder.blocks_enc." + std::to_string(i);
        expect_ln(c, b + ".norm1", C);
        expect_linear(c, b + ".attn.qkv", 3 * C, C);
        expect_linear(c, b + ".attn.proj", C, C);
        expect_ln(c, b + ".norm2", C);
        expect_linear(c, b + ".mlp.fc1", r * C, C);
        expect_linear(c, b + ".mlp.fc2", C, r * C);
    }
    expect_ln(c, "encoder.norm_enc", C);
    expect(c, "decoder.image2_embed", {1, 1, D});
    expect_linear(c, "decoder.feat_embed_enc_to_dec", D, C);
    for (int i = 0; i < g.dec_depth; ++i) {
        const std::string b = "decoder.blocks_dec." + std::to_string(i);
        expect_ln(c, b + ".norm1", D);
        expect_linear(c, b + ".attn.qkv", 3 * D, D);
        expect_linear(c, b + ".attn.proj", D, D);
        expect_ln(c, b + ".norm2", D);
        expect_ln(c, b + ".norm_y", D);
        expect_linear(c, b + ".cross_attn.projq", D, D);
        expect_linear(c, b + ".cross_attn.projk", D, D);
        expect_linear(c, b + ".cross_attn.projv", D, D);
        expect_linear(c, b + ".cross_attn.proj", D, D);
        expect_ln(c, b + ".norm3", D);
        expect_linear(c, b + ".mlp.fc1", r * D, D);
        expect_linear(c, b + ".mlp.fc2", D, r * D);
    }
    // feedback_mechanism.py:11-22: 'single_mlp' (fc1/fc2 + norm), 'single_linear' (weight/bias + norm) or no feedback layer:
    // all optional here, told apart by which keys were loaded; finalize rejects incomplete or mixed sets
    expect_linear(c, "decoder.feedback_layer.fc1", 4 * D, D, true);
    expect_linear(c, "decoder.feedback_layer.fc2", D, 4 * D, true);
    expect_linear(c, "decoder.feedback_layer", D, D, true);
    expect_ln(c, "decoder.feedback_norm", D, true);
    expect_ln(c, "decoder.norm_dec", D);
    expect_linear(c, "decoder.head_dec.proj", p * p * 7, D);
}

static Param* param(must3r_hip_ctx* c, const std::string& name) {
    auto it = c->params.find(name);
    return it == c->params.end() ? nullptr : &it->second;
}
static const float* p32(must3r_hip_ctx* c, const std::string& name) { return c->params.at(name).d; }

// packed 16-bit copy (and optional low part) of a parameter, created on first use for a dtype
static int p16p(must3r_hip_ctx* c, Param& p, DType dt, bool want_lo, const void** hi, const void** lo, hipStream_t s);
static int w16p(must3r_hip_ctx* c, Param& p, DType dt, const void** hi, hipStream_t s);
static int p16(must3r_hip_ctx* c, const std::string& name, DType dt, bool want_lo, const void** hi, const void** lo,
               hipStream_t s) {
    return p16p(c, c->params.at(name), dt, want_lo, hi, lo, s);
}
static int w16(must3r_hip_ctx* c, const std::string& name, DType dt, const void** hi, hipStream_t s) {
    return w16p(c, c->params.at(name), dt, hi, s);
}

// Param-based forms of p16 / w16 (the per-layer loops use the tables below instead of building the parameter name per launch)
static int p16p(must3r_hip_ctx* c, Param& p, DType dt, bool want_lo, const void** hi, const void** lo, hipStream_t s) {
    const char* err = "";
    if (!p.h16[dt] || (want_lo && !p.l16[dt])) {
        if (!p.h16[dt]) HIP_OK(hipMalloc(&p.h16[dt], p.n * 2));
        if (want_lo && !p.l16[dt]) HIP_OK(hipMalloc(&p.l16[dt], p.n * 2));
        if (launch_split16(dt, p.d, p.h16[dt], want_lo ? p.l16[dt] : nullptr, p.n, s, &err)) return fail("%s", err);
    }
    *hi = p.h16[dt];
    if (lo) *lo = p.l16[dt];
    return 0;
}
static int w16p(must3r_hip_ctx* c, Param& p, DType dt, const void** hi, hipStream_t s) {
    if (c->wsplit != 2) return p16p(c, p, dt, false, hi, nullptr, s);
    if (!p.x2[dt]) {
        const void *h, *l;
        M3R_OK(p16p(c, p, dt, true, &h, &l, s));
        const size_t rows = (size_t)p.shape[0], K = p.n / rows;
        HIP_OK(hipMalloc(&p.x2[dt], p.n * 4));
        HIP_OK(hipMemcpy2DAsync(p.x2[dt], K * 4, h, K * 2, K * 2, rows, hipMemcpyDeviceToDevice, s));
        HIP_OK(hipMemcpy2DAsync(reinterpret_cast<char*>(p.x2[dt]) + K * 2, K * 4, l, K * 2, K * 2, rows, hipMemcpyDeviceToDevice, s));
        // the separate hi / lo copies are dead once the interleaved rows exist (they are rebuilt from the fp32 master if a later call
        // asks for the plain 16-bit mode): 4 of the 16 bytes per parameter.  hipFree waits for the copies above (first use only).
        (void)hipFree(p.h16[dt]); p.h16[dt] = nullptr;
        (void)hipFree(p.l16[dt]); p.l16[dt] = nullptr;
    }
    // r05: the 2:4-sparse low part for the chip-filling kernels -- MUST3R_F16_WA only (mlp_plain): there the error budget is the plain Mlp weights' and the
    // dropped half of W_lo is invisible (scripts/emul/sparse_lo.py: 5.99e-4 vs 6.05e-4); MUST3R_F16_W2 exists for its 4e-4 and keeps the dense low part
    // (all-sparse would be 4.4e-4 against 3.4e-4).  fp16; M3R_SPARSE_LO=0: never.  rows x K / 2 values + rows x K / 8 bytes of positions, built on first use.
    const bool sparse_on = opt(OPT_SPARSE_LO) != 0;
    if (sparse_on && c->mlp_plain && dt == DT_F16 && !p.sp_vals) {
        const size_t rows = (size_t)p.shape[0], K = p.n / rows;
        if (rows % 128 == 0 && K % 64 == 0) {
            const char* err = "";
            HIP_OK(hipMalloc(&p.sp_vals, rows * K));
            HIP_OK(hipMalloc(&p.sp_idx, rows * K / 8));
            if (launch_sparse24_pack(p.d, (int)rows, (int)K, p.sp_vals, p.sp_idx, s, &err)) return fail("%s", err);
            c->sparse_lo[p.x2[dt]] = {p.sp_vals, p.sp_idx, (int)rows};
        }
    }
    *hi = p.x2[dt];
    return 0;
}
// weight operand of an Mlp Linear (fc1, fc2, the feedback Mlp): the plain 16-bit copy in MUST3R_F16_WA mode -- the launch then runs
// one MFMA pass (*ws = 0) -- and the context's layout otherwise
static int wmlp(must3r_hip_ctx* c, Param& p, DType dt, const void** w, int* ws, hipStream_t s) {
    if (c->wsplit == 2 && c->mlp_plain) { *ws = 0; return p16p(c, p, dt, false, w, nullptr, s); }
    *ws = c->wsplit;
    return w16p(c, p, dt, w, s);
}
static void build_layer_tables(must3r_hip_ctx* c, bool decoder) {
    std::vector<std::vector<Param*>>& tab = decoder ? c->dec_tab : c->enc_tab;
    const int depth = decoder ? c->cfg.dec_depth : c->cfg.enc_depth;
    tab.assign(depth, std::vector<Param*>(LF_COUNT, nullptr));
    for (int l = 0; l < depth; ++l) {
        const std::string b = (decoder ? "decoder.blocks_dec." : "encoder.blocks_enc.") + std::to_string(l) + ".";
        for (int f = 0; f < LF_COUNT; ++f) tab[l][f] = param(c, b + kLayerFieldNames[f]);
    }
}

// [W_hi | W_hi | W_lo] rows: with the activation laid out as [y_hi | y_lo | y_hi] the three products of the split-precision head
// (y_hi W_hi + y_lo W_hi + y_hi W_lo) are ONE plain GEMM over K = 3 in
static int w3(must3r_hip_ctx* c, const std::string& name, DType dt, const void** out, hipStream_t s) {
    Param& p = c->params.at(name);
    if (!p.x3[dt]) {
        const void *h, *l;
        M3R_OK(p16(c, name, dt, true, &h, &l, s));
        const size_t rows = (size_t)p.shape[0], K = p.n / rows;
        HIP_OK(hipMalloc(&p.x3[dt], p.n * 6));
        char* d = reinterpret_cast<char*>(p.x3[dt]);
        HIP_OK(hipMemcpy2DAsync(d, K * 6, h, K * 2, K * 2, rows, hipMemcpyDeviceToDevice, s));
        HIP_OK(hipMemcpy2DAsync(d + K * 2, K * 6, h, K * 2, K * 2, rows, hipMemcpyDeviceToDevice, s));
        HIP_OK(hipMemcpy2DAsync(d + K * 4, K * 6, l, K * 2, K * 2, rows, hipMemcpyDeviceToDevice, s));
    }
    *out = p.x3[dt];
    return 0;
}

extern "C" int must3r_hip_rope_table(float freq, float f0, int npos, float* out) {
    // angle(p, i) = p * f0 * freq^(-i/16), i in [0,16)  (croco RoPE2D, head dim 64)
    for (int p = 0; p < npos; ++p)
        for (int i = 0; i < 16; ++i) {
            const float inv = f0 / powf(freq, (float)i / 16.0f);
            const float ang = (float)p * inv;
            out[(p * 16 + i) * 2 + 0] = cosf(ang);
            out[(p * 16 + i) * 2 + 1] = sinf(ang);
        }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI: lifetime / weights
// ------------------------------------------------------------------------------------------------
extern "C" int must3r_hip_abi_version(void) { return MUST3R_HIP_ABI_VERSION; }
extern "C" int must3r_hip_set_option(const char* name, long long value) {
    const char* err = "";
    if (opt_set(name, value, &err)) return fail("%s", err);
    return 0;
}
extern "C" const char* must3r_hip_last_error(void) { return g_err; }

extern "C" int must3r_hip_create(const must3r_hip_config* cfg, int device, must3r_hip_ctx** out) {
    if (!cfg || !out) return fail("create: null argument");
    if (cfg->patch_size != 16) return fail("create: patch_size must be 16");
    if (cfg->enc_dim != cfg->enc_heads * 64 || cfg->dec_dim != cfg->dec_heads * 64)
        return fail("create: head dim must be 64 (enc %d/%d, dec %d/%d)", cfg->enc_dim, cfg->enc_heads, cfg->dec_dim, cfg->dec_heads);
    if (cfg->enc_dim % 64 || cfg->dec_dim % 64 || cfg->enc_dim > 1024 || cfg->dec_dim > 1024)
        return fail("create: feature dims must be multiples of 64 and <= 1024");
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail("create: no HIP device visible (the HIP path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail("create: bad device %d", device);
    DeviceGuard dev_guard(device);
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("create: device is %s, this library is built for gfx950 only", prop.gcnArchName);
    must3r_hip_ctx* c = new must3r_hip_ctx();
    c->cfg = *cfg;
    c->device = device;
    build_expected(c);
    if (hipHostMalloc(reinterpret_cast<void**>(&c->pin), must3r_hip_ctx::kSlots * must3r_hip_ctx::kSlotBytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->pin_dev), must3r_hip_ctx::kSlots * must3r_hip_ctx::kSlotBytes) != hipSuccess) {
        delete c;
        return fail("create: staging allocation failed");
    }
    for (int i = 0; i < must3r_hip_ctx::kSlots; ++i) {
        (void)hipEventCreateWithFlags(&c->slot_ev[i], hipEventDisableTiming);
        c->slot_used[i] = false;
    }
    *out = c;
    return 0;
}

extern "C" void must3r_hip_destroy(must3r_hip_ctx* c) {
    if (!c) return;
    DeviceGuard dev_guard(c->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : c->params) {
        Param& p = kv.second;
        if (p.d) (void)hipFree(p.d);
        for (int i = 0; i < 2; ++i) {
            if (p.h16[i]) (void)hipFree(p.h16[i]);
            if (p.l16[i]) (void)hipFree(p.l16[i]);
            if (p.x2[i]) (void)hipFree(p.x2[i]);
            if (p.x3[i]) (void)hipFree(p.x3[i]);
        }
        if (p.sp_vals) (void)hipFree(p.sp_vals);
        if (p.sp_idx) (void)hipFree(p.sp_idx);
    }
    if (c->rope_tab) (void)hipFree(c->rope_tab);
    if (c->ws) (void)hipFree(c->ws);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->pin_dev) (void)hipFree(c->pin_dev);
    for (int i = 0; i < must3r_hip_ctx::kSlots; ++i) (void)hipEventDestroy(c->slot_ev[i]);
    prof_flush(c);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    delete c;
}

extern "C" int must3r_hip_load_weight(must3r_hip_ctx* c, const char* name, const float* data, int is_device, int ndim,
                                      const int64_t* shape) {
    if (!c || !name || !data) return fail("load_weight: null argument");
    Param* p = param(c, name);
    if (!p || p->derived) return fail("load_weight: unexpected key '%s'", name);
    bool same = (int)p->shape.size() == ndim;
    for (int i = 0; same && i < ndim; ++i) same = p->shape[i] == shape[i];
    if (!same) return fail("load_weight: shape mismatch for '%s'", name);
    DeviceGuard dev_guard(c->device);
    if (!p->d) HIP_OK(hipMalloc(reinterpret_cast<void**>(&p->d), p->n * sizeof(float)));
    HIP_OK(hipMemcpy(p->d, data, p->n * sizeof(float), is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    for (int i = 0; i < 2; ++i) {  // invalidate packed copies
        if (p->h16[i]) { (void)hipFree(p->h16[i]); p->h16[i] = nullptr; }
        if (p->l16[i]) { (void)hipFree(p->l16[i]); p->l16[i] = nullptr; }
        if (p->x2[i]) { c->sparse_lo.erase(p->x2[i]); (void)hipFree(p->x2[i]); p->x2[i] = nullptr; }
        if (p->x3[i]) { (void)hipFree(p->x3[i]); p->x3[i] = nullptr; }
    }
    if (p->sp_vals) { (void)hipFree(p->sp_vals); p->sp_vals = nullptr; }
    if (p->sp_idx) { (void)hipFree(p->sp_idx); p->sp_idx = nullptr; }
    p->loaded = true;
    if (strncmp(name, "encoder.", 8) == 0) c->fin_enc = false; else c->fin_dec = false;
    return 0;
}

static int derive(must3r_hip_ctx* c, const std::string& name, std::vector<int64_t> shape, const std::vector<float>& host) {
    Param& p = c->params[name];
    if (p.d) { (void)hipFree(p.d); p.d = nullptr; }
    for (int i = 0; i < 2; ++i) {
        if (p.h16[i]) { (void)hipFree(p.h16[i]); p.h16[i] = nullptr; }
        if (p.l16[i]) { (void)hipFree(p.l16[i]); p.l16[i] = nullptr; }
        if (p.x2[i]) { c->sparse_lo.erase(p.x2[i]); (void)hipFree(p.x2[i]); p.x2[i] = nullptr; }
        if (p.x3[i]) { (void)hipFree(p.x3[i]); p.x3[i] = nullptr; }
    }
    if (p.sp_vals) { (void)hipFree(p.sp_vals); p.sp_vals = nullptr; }
    if (p.sp_idx) { (void)hipFree(p.sp_idx); p.sp_idx = nullptr; }
    p.shape = shape;
    p.n = host.size();
    p.derived = true;
    p.loaded = true;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&p.d), p.n * sizeof(float)));
    HIP_OK(hipMemcpy(p.d, host.data(), p.n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
static int fetch(must3r_hip_ctx* c, const std::string& name, std::vector<float>& host) {
    Param& p = c->params.at(name);
    host.resize(p.n);
    HIP_OK(hipMemcpy(host.data(), p.d, p.n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// "LN fold" operands of a Linear that follows a LayerNorm: W' = gamma (.) W, s_n = sum_k W'[n][k], c_n = (W beta + b)_n, so that
// LN(x) W^T + b = rstd (x W'^T - mu s) + c.  Derived parameters <block>.<linear>_ln.{weight, s, c}.
static int derive_ln_fold(must3r_hip_ctx* c, const std::string& pb, const char* norm, const char* lin) {
    std::vector<float> w, bb, gam, bet;
    M3R_OK(fetch(c, pb + lin + ".weight", w));
    M3R_OK(fetch(c, pb + lin + ".bias", bb));
    M3R_OK(fetch(c, pb + norm + ".weight", gam));
    M3R_OK(fetch(c, pb + norm + ".bias", bet));
    const size_t K = gam.size(), N = bb.size();
    if (w.size() != N * K || bet.size() != K) return fail("finalize: LN-fold shapes of %s%s", pb.c_str(), lin);
    std::vector<float> sn(N), cn(N);
    for (size_t n = 0; n < N; ++n) {
        double ss = 0.0, cc = bb[n];
        float* row = &w[n * K];
        for (size_t k = 0; k < K; ++k) {
            cc += (double)bet[k] * row[k];
            row[k] *= gam[k];
            ss += row[k];
        }
        sn[n] = (float)ss;
        cn[n] = (float)cc;
    }
    M3R_OK(derive(c, pb + lin + "_ln.weight", {(int64_t)N, (int64_t)K}, w));
    M3R_OK(derive(c, pb + lin + "_ln.s", {(int64_t)N}, sn));
    M3R_OK(derive(c, pb + lin + "_ln.c", {(int64_t)N}, cn));
    return 0;
}

extern "C" int must3r_hip_finalize_weights(must3r_hip_ctx* c, int parts) {
    if (!c) return fail("finalize: null context");
    if (!(parts & 3)) return fail("finalize: parts must include MUST3R_PART_ENCODER and/or MUST3R_PART_DECODER");
    DeviceGuard dev_guard(c->device);
    for (auto& kv : c->params) {
        const bool is_enc = kv.first.compare(0, 8, "encoder.") == 0;
        if (!((is_enc && (parts & 1)) || (!is_enc && (parts & 2)))) continue;
        if (!kv.second.loaded && !kv.second.derived && !kv.second.optional) return fail("finalize: missing key '%s'", kv.first.c_str());
    }
    if (parts & 2) {
        int mlp = 0, lin = 0, nrm = 0;
        for (const char* k : {"decoder.feedback_layer.fc1.weight", "decoder.feedback_layer.fc1.bias", "decoder.feedback_layer.fc2.weight",
                              "decoder.feedback_layer.fc2.bias"}) mlp += is_loaded(c, k);
        for (const char* k : {"decoder.feedback_layer.weight", "decoder.feedback_layer.bias"}) lin += is_loaded(c, k);
        for (const char* k : {"decoder.feedback_norm.weight", "decoder.feedback_norm.bias"}) nrm += is_loaded(c, k);
        const bool ok = (mlp == 4 && lin == 0 && nrm == 2) || (mlp == 0 && lin == 2 && nrm == 2) || (mlp == 0 && lin == 0 && nrm == 0);
        if (!ok) return fail("finalize: inconsistent feedback layer keys (single_mlp: fc1/fc2 + feedback_norm; single_linear: "
                             "weight/bias + feedback_norm; none: neither)");
    }
    // RoPE table for positions up to 256 (4096-pixel side), shared by both halves
    {
        c->rope_npos = 256;
        std::vector<float> t((size_t)c->rope_npos * 32);
        must3r_hip_rope_table(c->cfg.rope_freq, c->cfg.rope_f0, c->rope_npos, t.data());
        if (!c->rope_tab) HIP_OK(hipMalloc(reinterpret_cast<void**>(&c->rope_tab), t.size() * sizeof(float)));
        HIP_OK(hipMemcpy(c->rope_tab, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (parts & 1) {
        // r06: LN-fold operands of the encoder blocks (norm1 -> attn.qkv, norm2 -> mlp.fc1) for the chip-filling launches (GemmArgs::fold256)
        for (int l = 0; l < c->cfg.enc_depth; ++l) {
            const std::string pb = "encoder.blocks_enc." + std::to_string(l) + ".";
            M3R_OK(derive_ln_fold(c, pb, "norm1", "attn.qkv"));
            M3R_OK(derive_ln_fold(c, pb, "norm2", "mlp.fc1"));
        }
        c->enc_tab.clear();
        c->fin_enc = true;
    }
    if (!(parts & 2)) return 0;
    const must3r_hip_config& g = c->cfg;
    const int D = g.dec_dim;
    std::vector<float> a, b, o;
    // fused K|V projection per decoder layer: rows [0,D) = projk, [D,2D) = projv  (layers.py:87-88 cat([k,v],-1))
    for (int l = 0; l < g.dec_depth; ++l) {
        const std::string pb = "decoder.blocks_dec." + std::to_string(l) + ".cross_attn.";
        M3R_OK(fetch(c, pb + "projk.weight", a));
        M3R_OK(fetch(c, pb + "projv.weight", b));
        o = a; o.insert(o.end(), b.begin(), b.end());
        M3R_OK(derive(c, pb + "projkv.weight", {2 * D, D}, o));
        M3R_OK(fetch(c, pb + "projk.bias", a));
        M3R_OK(fetch(c, pb + "projv.bias", b));
        o = a; o.insert(o.end(), b.begin(), b.end());
        M3R_OK(derive(c, pb + "projkv.bias", {2 * D}, o));
    }
    // all layers' K|V projections and norm_y affine parameters stacked, for the grouped post-feedback projection
    {
        std::vector<float> wall, ball, nw, nb;
        for (int l = 0; l < g.dec_depth; ++l) {
            const std::string pb = "decoder.blocks_dec." + std::to_string(l);
            M3R_OK(fetch(c, pb + ".cross_attn.projkv.weight", a)); wall.insert(wall.end(), a.begin(), a.end());
            M3R_OK(fetch(c, pb + ".cross_attn.projkv.bias", a)); ball.insert(ball.end(), a.begin(), a.end());
            M3R_OK(fetch(c, pb + ".norm_y.weight", a)); nw.insert(nw.end(), a.begin(), a.end());
            M3R_OK(fetch(c, pb + ".norm_y.bias", a)); nb.insert(nb.end(), a.begin(), a.end());
        }
        M3R_OK(derive(c, "decoder.projkv_all.weight", {(int64_t)g.dec_depth * 2 * D, D}, wall));
        M3R_OK(derive(c, "decoder.projkv_all.bias", {(int64_t)g.dec_depth * 2 * D}, ball));
        M3R_OK(derive(c, "decoder.norm_y_all.weight", {(int64_t)g.dec_depth * D}, nw));
        M3R_OK(derive(c, "decoder.norm_y_all.bias", {(int64_t)g.dec_depth * D}, nb));
    }
    // pixel-shuffled head: new row (i*16+j)*7 + ch <- old row ch*256 + i*16 + j  (tools/image.py:9-14)
    {
        M3R_OK(fetch(c, "decoder.head_dec.proj.weight", a));
        M3R_OK(fetch(c, "decoder.head_dec.proj.bias", b));
        const int P = 256;
        std::vector<float> w2(a.size()), b2(b.size());
        for (int ch = 0; ch < 7; ++ch)
            for (int ij = 0; ij < P; ++ij) {
                const size_t nr = (size_t)ij * 7 + ch, orow = (size_t)ch * P + ij;
                memcpy(&w2[nr * D], &a[orow * D], sizeof(float) * D);
                b2[nr] = b[orow];
            }
        M3R_OK(derive(c, "decoder.head_dec.proj_ps.weight", {7 * P, D}, w2));
        M3R_OK(derive(c, "decoder.head_dec.proj_ps.bias", {7 * P}, b2));
    }
    // "LN fold" operands of the decoder blocks (one-view memory updates since r02, chip-filling batched calls since r06): the three Linears that follow a LayerNorm
    for (int l = 0; l < g.dec_depth; ++l) {
        const std::string pb = "decoder.blocks_dec." + std::to_string(l) + ".";
        M3R_OK(derive_ln_fold(c, pb, "norm1", "attn.qkv"));
        M3R_OK(derive_ln_fold(c, pb, "norm2", "cross_attn.projq"));
        M3R_OK(derive_ln_fold(c, pb, "norm3", "mlp.fc1"));
    }
    c->dec_tab.clear();   // rebuilt on the next decode (the derived entries above may be new)
    c->fin_dec = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// encoder  (Dust3rEncoder.forward encoder.py:46-52)
// ------------------------------------------------------------------------------------------------
static int encode_chunk(must3r_hip_ctx* c, DType dt, const float* img, int V, int H, int W, float* out_tokens,
                        int64_t* out_pos, hipStream_t s) {
    const must3r_hip_config& g = c->cfg;
    const int C = g.enc_dim, gh = H / 16, gw = W / 16, N = gh * gw, R = V * N, Hh = g.enc_heads, F = g.mlp_ratio * C;
    const char* err = "";
    const bool a8 = kAttnFp8Built && c->attn8 != 0;   // (compiles out of the default build, kernels.hpp)
    size_t need = 0;
    need = ws_need(need, (size_t)R * 768, 2);
    need = ws_need(need, (size_t)R * C, 4);
    need = ws_need(need, (size_t)R * C, 2);
    need = ws_need(need, (size_t)R * 3 * C, 2);
    need = ws_need(need, (size_t)R * C, 2);
    need = ws_need(need, (size_t)R * F, 2);
    need = ws_need(need, a8 ? (size_t)R * 2 * C : 0, 1);
    // r06, LN fold on the chip-filling launches (GemmArgs::fold256; MUST3R_F16_WA): norm1 of blocks 1.. and norm2 of every block are not launched -- the residual
    // GEMM in front of them (fc2 of the previous block / proj) leaves the shifted rows in fp16 + per-wave-tile sums, qkv / fc1 normalise after their product.
    // Block 0's norm1 keeps its kernel (its rows come from the patch embedding, an EPI_F32 launch).  Only when every Linear of the block lands on the fold kernels.
    if (c->enc_tab.empty()) build_layer_tables(c, false);
    const bool f256 = opt(OPT_LNFOLD256) != 0 && dt == DT_F16 && c->wsplit == 2 && c->mlp_plain && !a8 && (C == 768 || C == 1024) &&
                      c->enc_tab[0][LF_QKVLN_W] && c->enc_tab[0][LF_FC1LN_W] &&
                      gemm_fold256_shape_ok(R, 3 * C, C, true) && gemm_fold256_shape_ok(R, C, C, true) &&
                      gemm_fold256_shape_ok(R, F, C, false) && gemm_fold256_shape_ok(R, C, F, false);
    need = ws_need(need, f256 ? (size_t)R * C : 0, 2);                 // x16: the shifted residual rows in fp16
    need = ws_need(need, f256 ? (size_t)R * (C / 64) * 2 : 0, 4);      // per row and 64-column wave tile (sum, sum of squares)
    need = ws_need(need, f256 ? (size_t)R : 0, 4);                     // per row: the current mean (GemmArgs::ln_shift)
    M3R_OK(ws_reserve(c, need, s));
    uint16_t* P16 = ws_take<uint16_t>(c, (size_t)R * 768);
    float* x = ws_take<float>(c, (size_t)R * C);
    uint16_t* h16 = ws_take<uint16_t>(c, (size_t)R * C);
    uint16_t* qkv = ws_take<uint16_t>(c, (size_t)R * 3 * C);
    uint16_t* a16 = ws_take<uint16_t>(c, (size_t)R * C);
    uint16_t* g16 = ws_take<uint16_t>(c, (size_t)R * F);
    uint8_t* q8 = a8 ? ws_take<uint8_t>(c, (size_t)R * 2 * C) : nullptr;
    uint16_t* x16 = f256 ? ws_take<uint16_t>(c, (size_t)R * C) : nullptr;
    float* lnstats = f256 ? ws_take<float>(c, (size_t)R * (C / 64) * 2) : nullptr;
    float* lnshift = f256 ? ws_take<float>(c, (size_t)R) : nullptr;
    if (!g16 || (a8 && !q8) || (f256 && !lnshift)) return fail("encode: workspace sizing bug");
    if (f256) HIP_OK(hipMemsetAsync(lnshift, 0, (size_t)R * sizeof(float), s));
    auto fold_in = [&](GemmArgs& g_, const Param* sn, const Param* cn) {
        g_.A = x16; g_.ln_stats = lnstats; g_.ln_s = sn->d; g_.bias = cn->d; g_.ln_eps = 1e-6f; g_.ln_shift = lnshift; g_.fold256 = 1;
    };
    auto fold_out = [&](GemmArgs& g_) { g_.x16_out = x16; g_.stats_out = lnstats; g_.ln_shift = lnshift; g_.fold256 = 1; };

    {
        ProfScope ps(c, s, PC_MISC, 0.0);
        if (launch_fill_pos(out_pos, V, gh, gw, s, &err)) return fail("%s", err);
        if (launch_im2col(dt, img, P16, V, H, W, s, &err)) return fail("%s", err);
    }
    std::vector<AttnView> views(V);
    for (int v = 0; v < V; ++v) views[v] = AttnView{v * N, N, v * N, N, 0, 0};
    void* views_dev = nullptr;
    M3R_OK(upload_table(c, views.data(), sizeof(AttnView) * V, &views_dev, s));

    const void* w;
    M3R_OK(w16(c, "encoder.patch_embed.proj.weight", dt, &w, s));
    M3R_OK(gemm(c, dt, EPI_F32, gargs(P16, w, p32(c, "encoder.patch_embed.proj.bias"), x, R, C, 768, 768, C), s));
    for (int l = 0; l < g.enc_depth; ++l) {
        const std::vector<Param*>& LP = c->enc_tab[l];
        const bool fq = f256 && l > 0;
        if (!fq)
            M3R_OK(layernorm(c, dt, x, nullptr, LP[LF_N1W]->d, LP[LF_N1B]->d, h16, nullptr, nullptr,
                             nullptr, R, C, 1e-6f, s));
        M3R_OK(w16p(c, *LP[fq ? LF_QKVLN_W : LF_QKVW], dt, &w, s));
        GemmArgs ga = gargs(h16, w, LP[LF_QKVB]->d, qkv, R, 3 * C, C, C, 3 * C);
        if (fq) fold_in(ga, LP[LF_QKVLN_S], LP[LF_QKVLN_C]);
        ga.pos = out_pos; ga.rope_tab = c->rope_tab; ga.rope_cols = 2 * C; ga.rope_npos = c->rope_npos;
        ga.out_scale = kQScale; ga.scale_cols = C;   // q *= 1/sqrt(64) * log2(e)
        M3R_OK(gemm(c, dt, EPI_QKV_ROPE, ga, s));
        AttnArgs aa;
        memset(&aa, 0, sizeof(aa));
        aa.Q = qkv; aa.K = qkv + C; aa.V = qkv + 2 * C; aa.O = a16;
        aa.ldq = aa.ldk = aa.ldv = 3 * C; aa.ldo = C; aa.heads = Hh;
        aa.views = reinterpret_cast<const AttnView*>(views_dev); aa.nviews = V; aa.max_nq = N; aa.scale = 0.125f;
        aa.q_prescaled = 1;
        if (a8 && !attention_is_small(V, Hh, N, 1)) {   // e4m3 copies of q | k (one byte per element); V stays 16-bit
            M3R_OK(quant8(c, dt, qkv, 3 * C, q8, 2 * C, nullptr, 0, (size_t)R, 2 * C, 0, s));
            aa.Q = q8; aa.K = q8 + C; aa.ldq = aa.ldk = 2 * C; aa.fp8 = 1;
        }
        aa.max_nk = N;
        M3R_OK(attention(c, dt, aa, 4.0 * V * (double)N * N * C, PC_ATTN_SA, s));
        M3R_OK(w16p(c, *LP[LF_PROJW], dt, &w, s));
        {
            GemmArgs gp = gargs(a16, w, LP[LF_PROJB]->d, x, R, C, C, C, C);
            if (f256) fold_out(gp);
            M3R_OK(gemm(c, dt, EPI_RESID_F32, gp, s));
        }
        if (!f256)
            M3R_OK(layernorm(c, dt, x, nullptr, LP[LF_N2W]->d, LP[LF_N2B]->d, h16, nullptr, nullptr,
                             nullptr, R, C, 1e-6f, s));
        int ws;
        M3R_OK(wmlp(c, *LP[f256 ? LF_FC1LN_W : LF_FC1W], dt, &w, &ws, s));
        {
            GemmArgs g1 = gargs(h16, w, LP[LF_FC1B]->d, g16, R, F, C, C, F);
            if (f256) fold_in(g1, LP[LF_FC1LN_S], LP[LF_FC1LN_C]);
            M3R_OK(gemm(c, dt, EPI_STORE16_GELU, g1, s, ws));
        }
        M3R_OK(wmlp(c, *LP[LF_FC2W], dt, &w, &ws, s));
        {
            GemmArgs g2 = gargs(g16, w, LP[LF_FC2B]->d, x, R, C, F, F, C);
            if (f256 && l + 1 < g.enc_depth) fold_out(g2);   // the next block's norm1
            M3R_OK(gemm(c, dt, EPI_RESID_F32, g2, s, ws));
        }
    }
    M3R_OK(layernorm(c, dt, x, nullptr, p32(c, "encoder.norm_enc.weight"), p32(c, "encoder.norm_enc.bias"), nullptr, nullptr,
                     out_tokens, nullptr, R, C, 1e-6f, s));
    return 0;
}

extern "C" int must3r_hip_encode(must3r_hip_ctx* c, int dtype, const float* img, int n_views, int H, int W,
                                 float* out_tokens, int64_t* out_pos, void* stream) {
    if (!c || !img || !out_tokens || !out_pos) return fail("encode: null argument");
    if (!c->fin_enc) return fail("encode: encoder weights not finalized");
    if ((dtype & MUST3R_ATTN_FP8) && !kAttnFp8Built) return fail("encode: MUST3R_ATTN_FP8 is an experiment build (make EXTRA=-DM3R_ATTN_FP8): measured outside the 1e-3 target for +0.4 %%");
    c->attn8 = (dtype & MUST3R_ATTN_FP8) ? 1 : 0;
    dtype &= ~MUST3R_ATTN_FP8;
    if (dtype != MUST3R_BF16 && dtype != MUST3R_F16 && dtype != MUST3R_F16_W2 && dtype != MUST3R_F16_WA) return fail("encode: bad dtype %d", dtype);
    c->wsplit = (dtype == MUST3R_F16_W2 || dtype == MUST3R_F16_WA) ? 2 : 0;
    c->mlp_plain = dtype == MUST3R_F16_WA ? 1 : 0;
    if (dtype == MUST3R_F16_W2 || dtype == MUST3R_F16_WA) dtype = MUST3R_F16;
    if (n_views <= 0) return 0;
    if (H <= 0 || W <= 0 || H % 16 || W % 16) return fail("encode: H=%d W=%d must be positive multiples of 16", H, W);
    if (H / 16 > c->rope_npos || W / 16 > c->rope_npos) return fail("encode: image too large for the RoPE table");
    DeviceGuard dev_guard(c->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int N = (H / 16) * (W / 16);
    // rows per chunk: bounds the workspace (~32k token rows: the 16-bit activations of a chunk stay inside the 256 MB MALL between producer and
    // consumer launches).  M3R_ENC_CHUNK_ROWS: A/B instrument -- larger chunks lose less to the round quantisation of the
    // one-block-per-CU GEMMs (40 views: 94 % tile fill; 64 / 128 views: 100 %) but stream their activations through HBM.
    const int chunk_rows = opt(OPT_ENC_CHUNK_ROWS);
    int per = chunk_rows / N;
    if (per < 1) per = 1;
    {   // ... in chunks of equal size (80 views of 768 tokens: 2 x 40 -- 94 % tile fill -- instead of 42 + 38)
        const int nch = (n_views + per - 1) / per;
        per = (n_views + nch - 1) / nch;
    }
    for (int v0 = 0; v0 < n_views; v0 += per) {
        const int nv = (n_views - v0 < per) ? n_views - v0 : per;
        M3R_OK(encode_chunk(c, (DType)dtype, img + (size_t)v0 * 3 * H * W, nv, H, W, out_tokens + (size_t)v0 * N * c->cfg.enc_dim,
                            out_pos + (size_t)v0 * N * 2, s));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// decoder  (MUSt3R.forward_list decoder.py:158-265; forward :267-350 is the single-group case)
// ------------------------------------------------------------------------------------------------
// one native call = one launch sequence; must3r_hip_decode (below) validates the arguments and cuts oversize render calls
static int decode_impl(must3r_hip_ctx* c, const must3r_hip_decode_args* A, void* stream) {
    const int adt = A->dtype & ~MUST3R_ATTN_FP8;
    c->attn8 = (A->dtype & MUST3R_ATTN_FP8) ? 1 : 0;
    if (adt != MUST3R_BF16 && adt != MUST3R_F16 && adt != MUST3R_F16_W2 && adt != MUST3R_F16_WA) return fail("decode: bad dtype %d", A->dtype);
    c->wsplit = (adt == MUST3R_F16_W2 || adt == MUST3R_F16_WA) ? 2 : 0;
    c->mlp_plain = adt == MUST3R_F16_WA ? 1 : 0;
    if (A->mem_mode != MUST3R_MEM_KV && A->mem_mode != MUST3R_MEM_NORM_Y && A->mem_mode != MUST3R_MEM_RAW)
        return fail("decode: bad mem_mode %d", A->mem_mode);
    if (A->n_groups <= 0) return fail("decode: no input group");
    if (A->render && (A->first_call || A->n_mem <= 0)) return fail("decode: render needs a memory (decoder.py:278)");
    if (A->first_call && A->n_mem != 0) return fail("decode: first_call with a non-empty memory");
    DeviceGuard dev_guard(c->device);
    if (c->dec_tab.empty()) build_layer_tables(c, true);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const DType dt = (adt == MUST3R_F16_W2 || adt == MUST3R_F16_WA) ? DT_F16 : (DType)adt;
    const bool a8 = kAttnFp8Built && c->attn8 != 0;   // (the default build has no e4m3 attention path: every `a8` branch below compiles out)
    const must3r_hip_config& g = c->cfg;
    const int C = g.enc_dim, D = g.dec_dim, Hh = g.dec_heads, F = g.mlp_ratio * D, L = g.dec_depth, Nm = A->n_mem;
    const int OUT = g.patch_size * g.patch_size * 7;
    const char* err = "";

    // ---- row layout: scenes (batch elements, decoder.py:170-186: B), then groups, then views, then tokens (x_cat order of a scene,
    //      decoder.py:211-214).  The S scenes of a call never interact (memory, labels and attention are per batch element): they
    //      share every launch -- M = S x rows-per-scene in the GEMMs, S x views in the attention tables -- and nothing else.
    const int S = A->n_scenes > 1 ? A->n_scenes : 1;
    const long long mem_stride = S > 1 ? A->mem_scene_stride : 0;      // memory rows between two scenes' buffers
    int Rs = 0, views_s = 0, max_n = 0;
    std::vector<int> grow0(A->n_groups);
    for (int gi = 0; gi < A->n_groups; ++gi) {
        const must3r_hip_group& G = A->groups[gi];
        grow0[gi] = Rs;
        Rs += G.n_views * G.n_tokens;
        views_s += G.n_views;
        if (G.n_tokens > max_n) max_n = G.n_tokens;
    }
    const int R = S * Rs, total_views = S * views_s;
    const bool one_block = A->n_groups == 1;                           // a group's [S, n_views, n_tokens] rows ARE the call's row order
    const bool update = !A->render;
    // r06, context-parallel cross attention (include/must3r_hip.h must3r_hip_cp): Nm = THIS rank's shard of the memory (possibly empty), the call's semantics
    // follow the GLOBAL row count (validated by must3r_hip_decode: one-view update of one scene, 'kv' memory, 16-bit attention)
    const must3r_hip_cp* cp = A->cp;
    const int Nm_global = cp ? cp->n_mem_total : Nm;
    const bool use_mask = update && (Nm_global > 0 || views_s > 1);    // decoder.py:199 / :293
    const bool lone_view = update && views_s == 1 && Nm_global > 0;    // own tokens excluded -> keys = old memory only
    const bool need_pre_kv = update && !lone_view;                     // pre-feedback K|V of the new tokens are attended

    // ---- workspace
    size_t need = 0;
    need = ws_need(need, (size_t)R * C, 2);           // t16
    need = ws_need(need, (size_t)R * D, 4);           // x
    need = ws_need(need, (size_t)R * D, 2);           // h16
    need = ws_need(need, (size_t)R * 3 * D, 2);       // [y_hi | y_lo | y_hi] operand of the head
    need = ws_need(need, (size_t)R * 3 * D, 2);       // qkv
    need = ws_need(need, (size_t)R * D, 2);           // q16
    need = ws_need(need, (size_t)R * D, 2);           // a16
    need = ws_need(need, (size_t)R * F, 2);           // g16
    if (update) {
        need = ws_need(need, (size_t)L * R * D, 4);   // memorised layer inputs
        need = ws_need(need, (size_t)R * D, 4);       // feedback offset
        need = ws_need(need, (size_t)L * R * D, 2);   // norm_y of all layers (grouped K|V projection)
    }
    // LN fold (one-view update calls, fp16 + split weights): none of the 36 LayerNorm launches of the blocks is issued; the residual GEMMs
    // leave 16-bit rows + per-fragment sums, the Linears that follow normalise after their product (kernels.hpp GemmArgs "LN fold").
    const bool lnf_on = opt(OPT_LNFOLD) != 0;
    // (S > 1: the consumers of the fold only exist on the small-M tile shapes; a batched call is past the launch floor the fold removes)
    const bool lnf = lnf_on && update && !need_pre_kv && c->wsplit == 2 && dt == DT_F16 && !a8 && !A->feats && A->n_groups == 1 &&
                     S == 1 && D == 768 && F % 96 == 0;
    // r06: the same fold on the chip-filling launches of a batched call -- update or render -- (GemmArgs::fold256; MUST3R_F16_WA): block 0's norm1 keeps its kernel
    // (its rows come from an EPI_F32 launch), every other norm1 / norm2 / norm3 is folded.  Only when every Linear of the block lands on the fold kernels.
    const bool f256 = !lnf && opt(OPT_LNFOLD256) != 0 && c->wsplit == 2 && c->mlp_plain && dt == DT_F16 && !a8 && !A->feats && !A->cp && (D == 768 || D == 1024) &&
                      gemm_fold256_shape_ok(R, 3 * D, D, true) && gemm_fold256_shape_ok(R, D, D, true) &&
                      gemm_fold256_shape_ok(R, F, D, false) && gemm_fold256_shape_ok(R, D, F, false);
    const bool lnF = lnf || f256;
    need = ws_need(need, lnF ? (size_t)R * D : 0, 2);                 // x16: the residual stream rounded to fp16
    need = ws_need(need, lnF ? (size_t)R * (D / 16) * 2 : 0, 4);      // per row and 16-column fragment (f256: 64-column wave tile) (sum, sum of squares)
    need = ws_need(need, lnF ? (size_t)R : 0, 4);                     // per row: the mean the last consumer measured (shift of the next rows)
    // split-KV cross attention when the launch cannot fill the chip (sequential memory update: one view per call)
    const int max_nk_ca = A->render ? Nm : Nm + (lone_view ? 0 : Rs);
    // (context parallel: the local attention always leaves split-KV partials -- at least two splits, an empty one costs nothing -- for the partial merge)
    int ca_split = attention_pick_split(total_views, Hh, max_n, max_nk_ca);
    if (cp) ca_split = Nm > 0 ? (ca_split > 2 ? ca_split : 2) : 0;
    const size_t split_bytes = attention_split_scratch_bytes(ca_split, R, Hh);
    need = ws_need(need, split_bytes, 1);
    // memory_mode 'norm_y' / 'raw': the memory holds (normalised / raw) tokens, K|V of ALL attended rows are projected
    // per layer per call into scratch (the reference re-projects them per query view, layers.py:92-96)
    const int mode = A->mem_mode;
    const int memD = mode == MUST3R_MEM_KV ? 2 * D : D;
    // bytes of a memory row: 16-bit elements, except the K (e4m3, D bytes) | V (16-bit, 2 D bytes) rows of the fp8-attention mode
    const size_t memRB = (a8 && mode == MUST3R_MEM_KV) ? (size_t)3 * D : (size_t)memD * 2;
    const size_t kvs_rows1 = mode == MUST3R_MEM_KV ? 0 : (size_t)max_nk_ca;   // per scene
    const size_t kvs_rows = kvs_rows1 * S;
    need = ws_need(need, kvs_rows * 2 * D, 2);
    need = ws_need(need, mode == MUST3R_MEM_RAW ? kvs_rows * D : 0, 2);
    // fp8 attention: e4m3 copies of q|k|v (self) / q (cross), 16-bit staging of freshly projected K|V rows before they are
    // quantised into the memory, e4m3 copy of the per-call K|V projection of the 'norm_y' / 'raw' modes
    const size_t kv16_rows = (a8 && mode == MUST3R_MEM_KV && update) ? (size_t)L * R : 0;
    need = ws_need(need, a8 ? (size_t)R * 2 * D : 0, 1);
    need = ws_need(need, kv16_rows * 2 * D, 2);
    need = ws_need(need, a8 ? kvs_rows * D : 0, 1);
    M3R_OK(ws_reserve(c, need, s));
    uint16_t* t16 = ws_take<uint16_t>(c, (size_t)R * C);
    float* x = ws_take<float>(c, (size_t)R * D);
    uint16_t* h16 = ws_take<uint16_t>(c, (size_t)R * D);
    uint16_t* hcat = ws_take<uint16_t>(c, (size_t)R * 3 * D);
    uint16_t* qkv = ws_take<uint16_t>(c, (size_t)R * 3 * D);
    uint16_t* q16 = ws_take<uint16_t>(c, (size_t)R * D);
    uint16_t* a16 = ws_take<uint16_t>(c, (size_t)R * D);
    uint16_t* g16 = ws_take<uint16_t>(c, (size_t)R * F);
    float* newmem = update ? ws_take<float>(c, (size_t)L * R * D) : nullptr;
    float* off32 = update ? ws_take<float>(c, (size_t)R * D) : nullptr;
    uint16_t* yall = update ? ws_take<uint16_t>(c, (size_t)L * R * D) : nullptr;
    uint16_t* x16 = lnF ? ws_take<uint16_t>(c, (size_t)R * D) : nullptr;
    float* lnstats = lnF ? ws_take<float>(c, (size_t)R * (D / 16) * 2) : nullptr;
    float* lnshift = lnF ? ws_take<float>(c, (size_t)R) : nullptr;
    bool lnshift_fresh = true;   // nothing measured yet in this call: the first consumer starts the estimate
    if (f256) HIP_OK(hipMemsetAsync(lnshift, 0, (size_t)R * sizeof(float), s));   // (the fold256 consumers add to it; its first producer reads zeros)
    char* split_ws = split_bytes ? ws_take<char>(c, split_bytes) : nullptr;
    uint16_t* kvs = kvs_rows ? ws_take<uint16_t>(c, kvs_rows * 2 * D) : nullptr;
    uint16_t* ytmp = (mode == MUST3R_MEM_RAW && kvs_rows) ? ws_take<uint16_t>(c, kvs_rows * D) : nullptr;
    uint8_t* q8 = a8 ? ws_take<uint8_t>(c, (size_t)R * 2 * D) : nullptr;
    uint16_t* kv16 = kv16_rows ? ws_take<uint16_t>(c, kv16_rows * 2 * D) : nullptr;
    uint8_t* kvs8 = (a8 && kvs_rows) ? ws_take<uint8_t>(c, kvs_rows * D) : nullptr;   // e4m3 K of the per-call projection ('norm_y' / 'raw')
    if (!g16 || (update && !off32) || (split_bytes && !split_ws) || (a8 && !q8) || (kv16_rows && !kv16) || (a8 && kvs_rows && !kvs8))
        return fail("decode: workspace sizing bug");

    // ---- per-view tables: self-attention, cross-attention; positions gathered into one [R,2] array
    std::vector<AttnView> tab(2 * (size_t)total_views);
    double sa_flops = 0, ca_flops = 0;
    {
        // K|V rows a scene's cross attention reads: its own memory buffer ('kv': rows b * mem_stride of the layer's buffer) or its slice
        // of the per-call projection scratch ('norm_y' / 'raw')
        const long long kv_stride = mode == MUST3R_MEM_KV ? mem_stride : (long long)kvs_rows1;
        int vi = 0;
        for (int b = 0; b < S; ++b)
            for (int gi = 0; gi < A->n_groups; ++gi) {
                const must3r_hip_group& G = A->groups[gi];
                for (int j = 0; j < G.n_views; ++j, ++vi) {
                    const int rl = grow0[gi] + j * G.n_tokens, n = G.n_tokens;   // row inside the scene
                    const int r0 = b * Rs + rl;
                    tab[vi] = AttnView{r0, n, r0, n, 0, 0};
                    AttnView cv{r0, n, (int)(b * kv_stride), Nm, 0, 0};
                    if (update) {
                        if (lone_view) { cv.nk = Nm; }
                        else if (use_mask && A->causal) {
                            // CausalMUSt3R.make_attn_mask (decoder.py:389-433): old memory + the new tokens of the views before this one = a PREFIX of the key rows;
                            // empty memory: view 0 attends view 1's tokens (labels < 2, != 0: decoder.py:399-402) = rows [n, 2n) behind an excluded [0, n)
                            if (Nm == 0 && j == 0) { cv.nk = 2 * n; cv.skip_lo = 0; cv.skip_hi = n; }
                            else { cv.nk = Nm + rl; }
                        }
                        else if (use_mask) { cv.nk = Nm + Rs; cv.skip_lo = Nm + rl; cv.skip_hi = Nm + rl + n; }
                        else { cv.nk = Nm + Rs; }  // first view alone: attends its own (pre-feedback) tokens
                    }
                    tab[total_views + vi] = cv;
                    sa_flops += 4.0 * n * (double)n * D;
                    ca_flops += 4.0 * n * (double)(cv.nk - (cv.skip_hi - cv.skip_lo)) * D;
                }
            }
    }
    void* tab_dev = nullptr;
    M3R_OK(upload_table(c, tab.data(), sizeof(AttnView) * tab.size(), &tab_dev, s));
    const AttnView* sa_views = reinterpret_cast<const AttnView*>(tab_dev);
    const AttnView* ca_views = sa_views + total_views;
    const int64_t* pos_all = A->groups[0].pos;  // single group: the caller's array is already [R,2]

    // per-(layer, scene) destinations of freshly projected K|V rows: memory rows [Nm, Nm + Rs) of scene b in layer l's buffer;
    // second half: the 16-bit staging rows the fp8 memory is quantised from
    void* kvout_dev = nullptr;
    if (update && mode == MUST3R_MEM_KV) {
        std::vector<void*> outs(2 * (size_t)L * S);
        for (int l = 0; l < L; ++l)
            for (int b = 0; b < S; ++b) {
                void* memrow = reinterpret_cast<char*>(A->mem[l]) + ((size_t)b * mem_stride + Nm) * memRB;
                outs[(size_t)l * S + b] = a8 ? static_cast<void*>(kv16 + ((size_t)l * S + b) * Rs * 2 * D) : memrow;
                outs[(size_t)L * S + (size_t)l * S + b] = memrow;
            }
        M3R_OK(upload_table(c, outs.data(), sizeof(void*) * outs.size(), &kvout_dev, s));
    }
    void* const* kvout = reinterpret_cast<void* const*>(kvout_dev);

    // ---- tokens -> 16 bit, enc->dec projection + image2_embed  (decoder.py:172-181 / :275-287)
    for (int b = 0; b < (one_block ? 1 : S); ++b)
        for (int gi = 0; gi < A->n_groups; ++gi) {
            const must3r_hip_group& G = A->groups[gi];
            const size_t rg = (size_t)G.n_views * G.n_tokens;   // rows of this group per scene
            ProfScope ps(c, s, PC_MISC, 0.0);
            if (launch_cast(dt, G.tokens + (size_t)b * rg * C, t16 + ((size_t)b * Rs + grow0[gi]) * C, nullptr, (one_block ? S : 1) * rg * C, s, &err))
                return fail("%s", err);
        }
    const void* w;
    {
        M3R_OK(w16(c, "decoder.feat_embed_enc_to_dec.weight", dt, &w, s));
        GemmArgs ga = gargs(t16, w, p32(c, "decoder.feat_embed_enc_to_dec.bias"), x, R, D, C, C, D);
        ga.bias2 = p32(c, "decoder.image2_embed");
        ga.row_start2 = A->first_call ? A->groups[0].n_tokens : 0;  // reference view (group 0, view 0) of every scene gets no embed
        ga.row_period2 = S > 1 ? Rs : 0;
        if (lnf) { ga.x16_out = x16; ga.stats_out = lnstats; ga.copy32_out = newmem; }   // block 0's norm1 input (+ its memorised copy)
        M3R_OK(gemm(c, dt, EPI_F32, ga, s));
    }
    // several aspect ratios: gather the positions of all rows into one [R,2] array.  t16 is dead after the
    // projection above (stream order) and is large enough (R*C*2 >= R*16 bytes).
    if (!one_block) {
        int64_t* pos_ws = reinterpret_cast<int64_t*>(t16);
        for (int b = 0; b < S; ++b)
            for (int gi = 0; gi < A->n_groups; ++gi) {
                const must3r_hip_group& G = A->groups[gi];
                const size_t rg = (size_t)G.n_views * G.n_tokens;
                HIP_OK(hipMemcpyAsync(pos_ws + ((size_t)b * Rs + grow0[gi]) * 2, G.pos + (size_t)b * rg * 2, rg * 16, hipMemcpyDeviceToDevice, s));
            }
        pos_all = pos_ws;
    }

    // prepare_y (layers.py:81-88) of the R new token rows, written to memory rows [Nm, Nm+R) in the caller's mode:
    //   'kv'     LN(norm_y) -> [projk | projv]       'norm_y'  LN(norm_y)        'raw'  the tokens themselves
    auto kv_project = [&](int l, const float* src, const float* add, float* copy, hipStream_t st) -> int {
        const std::vector<Param*>& LP = c->dec_tab[l];
        if (mode != MUST3R_MEM_KV) {   // the LayerNorm itself writes the scene's memory rows (one launch per scene)
            for (int b = 0; b < S; ++b) {
                uint16_t* dst = reinterpret_cast<uint16_t*>(A->mem[l]) + ((size_t)b * mem_stride + Nm) * memD;
                const size_t o = (size_t)b * Rs * D;
                LnArgs la = lnargs(src + o, add ? add + o : nullptr, LP[LF_NYW]->d, LP[LF_NYB]->d, nullptr, nullptr, nullptr, copy ? copy + o : nullptr,
                                   Rs, D, 1e-6f);
                if (mode == MUST3R_MEM_NORM_Y) la.out16 = dst; else la.raw16 = dst;
                M3R_OK(layernorm_a(c, dt, la, st));
            }
            return 0;
        }
        M3R_OK(layernorm_a(c, dt, lnargs(src, add, LP[LF_NYW]->d, LP[LF_NYB]->d, h16, nullptr, nullptr, copy, R, D, 1e-6f), st));
        const void* wk;
        M3R_OK(w16p(c, *LP[LF_PKVW], dt, &wk, st));
        // one problem per scene (same weights): rows [b Rs, (b+1) Rs) -> memory rows of scene b (fp8 memory: 16-bit staging rows first)
        GemmArgs gk = gargs(h16, wk, LP[LF_PKVB]->d, nullptr, Rs, 2 * D, D, D, 2 * D);
        gk.batch = S; gk.strideA = (long long)Rs * D; gk.out_table = kvout + (size_t)l * S;
        if (S == 1) { gk.batch = 0; gk.out_table = nullptr; gk.out = a8 ? static_cast<void*>(kv16) : reinterpret_cast<char*>(A->mem[l]) + (size_t)Nm * memRB; }
        M3R_OK(gemm(c, dt, EPI_STORE16, gk, st));
        if (!a8) return 0;
        // K -> e4m3, V copied: memory rows [K e4m3 | V 16-bit]
        return quant8(c, dt, kv16 + (size_t)l * S * Rs * 2 * D * (S > 1 ? 1 : 0), 2 * D, S == 1 ? reinterpret_cast<char*>(A->mem[l]) + (size_t)Nm * memRB : nullptr,
                      3 * D, S > 1 ? kvout + (size_t)L * S + (size_t)l * S : nullptr, Rs, (size_t)R, D, D, st);
    };
    // K|V rows the cross attention of layer l reads: the memory itself ('kv') or a projection of it into scratch
    auto kv_source = [&](int l, const void** kptr, hipStream_t st) -> int {
        if (mode == MUST3R_MEM_KV) { *kptr = A->mem[l]; return 0; }
        const std::vector<Param*>& LP = c->dec_tab[l];
        const int rows = (int)kvs_rows1;
        const void* wk;
        M3R_OK(w16p(c, *LP[LF_PKVW], dt, &wk, st));
        for (int b = 0; b < S; ++b) {
            const uint16_t* src = reinterpret_cast<const uint16_t*>(A->mem[l]) + (size_t)b * mem_stride * D;
            if (mode == MUST3R_MEM_RAW) {  // y_ = norm_y(y) on the stored tokens (layers.py:92)
                LnArgs la = lnargs(nullptr, nullptr, LP[LF_NYW]->d, LP[LF_NYB]->d, ytmp, nullptr, nullptr, nullptr,
                                   rows, D, 1e-6f);
                la.x16 = src;
                M3R_OK(layernorm_a(c, dt, la, st));
                src = ytmp;
            }
            M3R_OK(gemm(c, dt, EPI_STORE16, gargs(src, wk, LP[LF_PKVB]->d, kvs + (size_t)b * rows * 2 * D, rows, 2 * D, D, D, 2 * D), st));
        }
        *kptr = kvs;
        if (a8) M3R_OK(quant8(c, dt, kvs, 2 * D, kvs8, D, nullptr, 0, (size_t)kvs_rows, D, 0, st));   // K -> e4m3 (V is read from kvs)
        return 0;
    };

    for (int l = 0; l < L; ++l) {
        const std::vector<Param*>& LP = c->dec_tab[l];
        if (need_pre_kv) M3R_OK(kv_project(l, x, nullptr, nullptr, s));
        // --- self attention (layers.py:91)
        // LN fold: a consumer reads the raw 16-bit rows + fragment sums its producer left, with the gamma-scaled weight
        auto fold_in = [&](GemmArgs& g_, int fs) {
            g_.A = x16; g_.ln_stats = lnstats; g_.ln_s = LP[fs]->d; g_.bias = LP[fs + 1]->d; g_.ln_eps = 1e-6f;
            g_.ln_shift = lnshift; g_.ln_shift_init = (lnshift_fresh && !f256) ? 1 : 0;   // (f256: the buffer was zeroed)
            g_.fold256 = f256 ? 1 : 0;
            lnshift_fresh = false;
        };
        // ... and a residual GEMM leaves them for the next consumer (copy: the memorised input of the next block, decoder.py:304-305)
        auto fold_out = [&](GemmArgs& g_, float* copy) {
            g_.x16_out = x16; g_.stats_out = lnstats; g_.copy32_out = copy; g_.ln_shift = lnshift; g_.fold256 = f256 ? 1 : 0;
        };
        const bool fq = lnf || (f256 && l > 0);   // is this block's norm1 folded?
        if (!fq)
            M3R_OK(layernorm_a(c, dt, lnargs(x, nullptr, LP[LF_N1W]->d, LP[LF_N1B]->d, h16, nullptr, nullptr,
                                            update ? newmem + (size_t)l * R * D : nullptr, R, D, 1e-6f), s));
        M3R_OK(w16p(c, *LP[fq ? LF_QKVLN_W : LF_QKVW], dt, &w, s));
        GemmArgs ga = gargs(h16, w, LP[LF_QKVB]->d, qkv, R, 3 * D, D, D, 3 * D);
        if (fq) fold_in(ga, LF_QKVLN_S);
        ga.pos = pos_all; ga.rope_tab = c->rope_tab; ga.rope_cols = 2 * D; ga.rope_npos = c->rope_npos;
        ga.out_scale = kQScale; ga.scale_cols = D;
        M3R_OK(gemm(c, dt, EPI_QKV_ROPE, ga, s));
        AttnArgs aa;
        memset(&aa, 0, sizeof(aa));
        aa.Q = qkv; aa.K = qkv + D; aa.V = qkv + 2 * D; aa.O = a16;
        aa.ldq = aa.ldk = aa.ldv = 3 * D; aa.ldo = D; aa.heads = Hh;
        aa.views = sa_views; aa.nviews = total_views; aa.max_nq = max_n; aa.scale = 0.125f; aa.q_prescaled = 1;
        if (total_views == 1) { aa.view0_inline = 1; aa.view0 = tab[0]; }
        if (a8 && !attention_is_small(total_views, Hh, max_n, 1)) {
            M3R_OK(quant8(c, dt, qkv, 3 * D, q8, 2 * D, nullptr, 0, (size_t)R, 2 * D, 0, s));
            aa.Q = q8; aa.K = q8 + D; aa.ldq = aa.ldk = 2 * D; aa.fp8 = 1;
        }
        aa.max_nk = max_n;
        M3R_OK(attention(c, dt, aa, sa_flops, PC_ATTN_SA, s));
        M3R_OK(w16p(c, *LP[LF_PROJW], dt, &w, s));
        {
            GemmArgs gp = gargs(a16, w, LP[LF_PROJB]->d, x, R, D, D, D, D);
            if (lnF) fold_out(gp, nullptr);
            M3R_OK(gemm(c, dt, EPI_RESID_F32, gp, s));
        }
        // --- cross attention over the memory (layers.py:92-97; attention.py:139-149)
        if (!lnF)
            M3R_OK(layernorm(c, dt, x, nullptr, LP[LF_N2W]->d, LP[LF_N2B]->d, h16, nullptr, nullptr, nullptr,
                             R, D, 1e-6f, s));
        M3R_OK(w16p(c, *LP[lnF ? LF_PQLN_W : LF_PQW], dt, &w, s));
        {
            GemmArgs gq = gargs(h16, w, LP[LF_PQB]->d, q16, R, D, D, D, D);
            if (lnF) fold_in(gq, LF_PQLN_S);
            gq.out_scale = kQScale; gq.scale_cols = D;
            M3R_OK(gemm(c, dt, EPI_STORE16, gq, s));
        }
        const void* mk = nullptr;
        M3R_OK(kv_source(l, &mk, s));
        memset(&aa, 0, sizeof(aa));
        aa.Q = q16; aa.K = mk; aa.V = reinterpret_cast<const uint16_t*>(mk) + D; aa.O = a16;
        aa.ldq = D; aa.ldk = aa.ldv = 2 * D; aa.ldo = D; aa.heads = Hh;
        if (a8) {   // e4m3 q and K; 'kv' memory rows are [K e4m3 (D bytes) | V 16-bit]: ldk in bytes, ldv in 16-bit elements
            M3R_OK(quant8(c, dt, q16, D, q8, D, nullptr, 0, (size_t)R, D, 0, s));
            aa.Q = q8; aa.fp8 = 1;
            if (mode == MUST3R_MEM_KV) { aa.ldk = 3 * D; aa.ldv = 3 * D / 2; aa.V = reinterpret_cast<const char*>(mk) + D; }
            else { aa.K = kvs8; aa.ldk = D; }   // per-call projection: K from its e4m3 copy, V from the 16-bit scratch rows
        }
        aa.max_nk = max_nk_ca;
        aa.views = ca_views; aa.nviews = total_views; aa.max_nq = max_n; aa.scale = 0.125f; aa.q_prescaled = 1;
        if (total_views == 1) { aa.view0_inline = 1; aa.view0 = tab[total_views]; }
        if (ca_split > 1) {
            aa.nsplit = ca_split; aa.total_q_rows = R; aa.dense_rows = 1;
            aa.part_o = reinterpret_cast<float*>(split_ws);
            aa.part_ml = aa.part_o + (size_t)ca_split * R * D;
        }
        if (cp) {
            // this rank's keys -> ONE partial in its slot (fp32 un-normalised, or 16-bit normalised with must3r_hip_cp::partial16); all-gather of the slots (the
            // caller's collective); merge of the world's partials into a16
            const int p16 = cp->partial16 ? 1 : 0;
            char* const slot0 = reinterpret_cast<char*>(cp->slots);
            const size_t o_bytes = (size_t)R * D * (p16 ? 2 : 4);
            float* const my_o = reinterpret_cast<float*>(slot0 + (size_t)cp->rank * cp->slot_bytes);
            float* const my_ml = reinterpret_cast<float*>(slot0 + (size_t)cp->rank * cp->slot_bytes + o_bytes);
            if (Nm > 0) {
                {
                    ProfScope ps(c, s, PC_ATTN_CA, ca_flops);
                    if (launch_attention_phase(dt, aa, 1, s, &err)) return fail("%s", err);
                    ps.kernel(attention_last_kernel(), "/cross");
                }
                ProfScope ps(c, s, PC_ATTN_COMBINE, 0.0);
                aa.total_q_rows = R;
                if (launch_attention_partial_merge(dt, aa, my_o, my_ml, p16, s, &err)) return fail("%s", err);
            } else {
                ProfScope ps(c, s, PC_ATTN_COMBINE, 0.0);
                if (launch_attention_partial_empty(my_o, my_ml, R, Hh, p16, s, &err)) return fail("%s", err);
            }
            if (cp->exchange(cp->user, l, cp->slots, cp->slot_bytes, cp->world, cp->rank, stream))
                return fail("decode: the context-parallel exchange of layer %d failed", l);
            ProfScope ps(c, s, PC_ATTN_COMBINE, 0.0);
            aa.total_q_rows = R;
            if (launch_attention_partial_final(dt, aa, reinterpret_cast<const float*>(slot0), reinterpret_cast<const float*>(slot0 + o_bytes),
                                               (long long)(cp->slot_bytes / (p16 ? 2 : 4)), (long long)(cp->slot_bytes / 4), cp->world, p16, s, &err)) return fail("%s", err);
        } else
        M3R_OK(attention(c, dt, aa, ca_flops, PC_ATTN_CA, s));
        M3R_OK(w16p(c, *LP[LF_CPW], dt, &w, s));
        {
            GemmArgs gp = gargs(a16, w, LP[LF_CPB]->d, x, R, D, D, D, D);
            if (lnF) fold_out(gp, nullptr);
            M3R_OK(gemm(c, dt, EPI_RESID_F32, gp, s));
        }
        // --- MLP (layers.py:98)
        if (!lnF)
            M3R_OK(layernorm(c, dt, x, nullptr, LP[LF_N3W]->d, LP[LF_N3B]->d, h16, nullptr, nullptr, nullptr,
                             R, D, 1e-6f, s));
        int ws_mlp;
        M3R_OK(wmlp(c, *LP[lnF ? LF_FC1LN_W : LF_FC1W], dt, &w, &ws_mlp, s));
        {
            GemmArgs g1 = gargs(h16, w, LP[LF_FC1B]->d, g16, R, F, D, D, F);
            if (lnF) fold_in(g1, LF_FC1LN_S);
            M3R_OK(gemm(c, dt, EPI_STORE16_GELU, g1, s, ws_mlp));
        }
        M3R_OK(wmlp(c, *LP[LF_FC2W], dt, &w, &ws_mlp, s));
        {
            GemmArgs g2 = gargs(g16, w, LP[LF_FC2B]->d, x, R, D, F, F, D);
            if (lnF && l + 1 < L) fold_out(g2, update ? newmem + (size_t)(l + 1) * R * D : nullptr);   // the next block's norm1 input (+ its memorised copy)
            M3R_OK(gemm(c, dt, EPI_RESID_F32, g2, s, ws_mlp));
        }
        if (A->feats && l < L - 1)   // return_feats: the residual stream after block l (decoder.py:321)
            HIP_OK(hipMemcpyAsync(A->feats + (size_t)l * R * D, x, (size_t)R * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    }

    // (Measured and rejected in r02: forking this head onto a second stream so that it overlaps the feedback / memory-write chain that ends
    // an update call -- the two chains are independent -- costs 2.5 ms per scene instead of saving 0.5: the head's blocks take CUs from
    // the latency-critical chain.  273 vs 282 views/s, interleaved runs on one box.)
    hipStream_t hs_ = s;
    {
    // --- prediction head in split precision (fp32-equivalent; decoder.py:149-156 runs it in fp32):
        //     y = LN(x); out = y_hi W_hi + y_lo W_hi + y_hi W_lo + b, pixel-shuffled to [n,H,W,7]
        {
            LnArgs la = lnargs(x, nullptr, p32(c, "decoder.norm_dec.weight"), p32(c, "decoder.norm_dec.bias"), hcat, hcat + D,
                               A->feats ? A->feats + (size_t)(L - 1) * R * D : nullptr, nullptr, R, D, 1e-6f);
            la.out16_dup = hcat + 2 * D;
            la.ld16 = 3 * D;
            M3R_OK(layernorm_a(c, dt, la, hs_));
        }
        const void* wcat;
        M3R_OK(w3(c, "decoder.head_dec.proj_ps.weight", dt, &wcat, hs_));
        for (int b = 0; b < (one_block ? 1 : S); ++b)
            for (int gi = 0; gi < A->n_groups; ++gi) {
                const must3r_hip_group& G = A->groups[gi];
                const int rg = G.n_views * G.n_tokens;                       // rows of this group per scene
                const int Rg = one_block ? S * rg : rg;                      // one group: all scenes in one launch ([S, n, H, W, 7] is contiguous)
                const size_t scene_elems = (size_t)G.n_views * G.H * G.W * 7;
                const size_t scene_stride = G.pointmaps_scene_stride > 0 ? (size_t)G.pointmaps_scene_stride : scene_elems;
                GemmArgs ga = gargs(hcat + ((size_t)b * Rs + grow0[gi]) * 3 * D, wcat, p32(c, "decoder.head_dec.proj_ps.bias"),
                                    G.pointmaps + (size_t)b * scene_stride, Rg, OUT, 3 * D, 3 * D, 0);
                ga.ntok = G.n_tokens; ga.gw = G.W / 16; ga.H = G.H; ga.Wimg = G.W;
                if (one_block && scene_stride != scene_elems) { ga.head_views = G.n_views; ga.head_scene_skip = (long long)(scene_stride - scene_elems); }
                M3R_OK(gemm(c, dt, EPI_HEAD, ga, hs_));
            }
    }
    if (update) {
        // --- feedback (feedback_mechanism.py:39-53): offset = layer(LN_1e-5(new_mem[L-1])), added to layers 0..L-2;
        //     layer = Mlp ('single_mlp'), Linear ('single_linear') or nothing (feedback_type None: no offset)
        const bool fb_mlp = is_loaded(c, "decoder.feedback_layer.fc1.weight");
        const bool fb_lin = !fb_mlp && is_loaded(c, "decoder.feedback_layer.weight");
        if (fb_mlp || fb_lin)
            M3R_OK(layernorm(c, dt, newmem + (size_t)(L - 1) * R * D, nullptr, p32(c, "decoder.feedback_norm.weight"),
                             p32(c, "decoder.feedback_norm.bias"), h16, nullptr, nullptr, nullptr, R, D, 1e-5f, s));
        if (fb_mlp) {
            int ws_fb;
            M3R_OK(wmlp(c, c->params.at("decoder.feedback_layer.fc1.weight"), dt, &w, &ws_fb, s));
            M3R_OK(gemm(c, dt, EPI_STORE16_GELU, gargs(h16, w, p32(c, "decoder.feedback_layer.fc1.bias"), g16, R, 4 * D, D, D, 4 * D), s, ws_fb));
            M3R_OK(wmlp(c, c->params.at("decoder.feedback_layer.fc2.weight"), dt, &w, &ws_fb, s));
            M3R_OK(gemm(c, dt, EPI_F32, gargs(g16, w, p32(c, "decoder.feedback_layer.fc2.bias"), off32, R, D, 4 * D, 4 * D, D), s, ws_fb));
        } else if (fb_lin) {
            M3R_OK(w16(c, "decoder.feedback_layer.weight", dt, &w, s));
            M3R_OK(gemm(c, dt, EPI_F32, gargs(h16, w, p32(c, "decoder.feedback_layer.bias"), off32, R, D, D, D, D), s));
        } else {
            off32 = nullptr;
        }
        // --- stored memory = prepare_y(new_mem + offset) (decoder.py:236-239 / :327-330)
        if (mode == MUST3R_MEM_KV) {
            // the L projections are independent: ONE grouped LayerNorm over [L*R, D] (per-layer affine, offset on layers
            // < L-1) and ONE grouped GEMM writing each layer's K|V rows into its own memory buffer
            LnArgs la = lnargs(newmem, off32, p32(c, "decoder.norm_y_all.weight"), p32(c, "decoder.norm_y_all.bias"), yall, nullptr,
                               nullptr, nullptr, L * R, D, 1e-6f);
            la.rows_per_group = R; la.add_groups = L - 1;
            M3R_OK(layernorm_a(c, dt, la, s));
            // fp8 memory: the GEMM writes the 16-bit staging rows [L][S][Rs][2D], one grouped quantisation scatters them into the
            // layers' memory rows.  Problem g = l * S + b: rows of scene b, weights of layer l = g / S.
            const void* wk;
            M3R_OK(w16(c, "decoder.projkv_all.weight", dt, &wk, s));
            GemmArgs gk = gargs(yall, wk, p32(c, "decoder.projkv_all.bias"), nullptr, Rs, 2 * D, D, D, 2 * D);
            gk.batch = L * S; gk.wdiv = S; gk.strideA = (long long)Rs * D; gk.strideW = (long long)2 * D * D * (c->wsplit == 2 ? 2 : 1);
            gk.strideB = 2 * D; gk.out_table = kvout;
            M3R_OK(gemm(c, dt, EPI_STORE16, gk, s));
            if (a8) M3R_OK(quant8(c, dt, kv16, 2 * D, nullptr, 3 * D, kvout + (size_t)L * S, Rs, (size_t)L * R, D, D, s));
        } else {
            for (int l = 0; l < L; ++l)
                M3R_OK(kv_project(l, newmem + (size_t)l * R * D, l < L - 1 ? off32 : nullptr, nullptr, s));
        }
    }

    return 0;
}

// MUSt3R.forward / forward_list (decoder.py:158-350).  Validates everything the launches rely on (so that a non-Python caller
// cannot overrun a memory buffer or a table) and cuts render calls whose view tables would not fit the staging slot: rendered
// views (and scenes) are independent of each other, the pieces give the same pointmaps.
extern "C" int must3r_hip_has_fp8_attention(void) { return kAttnFp8Built ? 1 : 0; }

static size_t cp_slot_bytes(const must3r_hip_ctx* c, int rows, int partial16) {
    return align_up((size_t)rows * ((size_t)c->cfg.dec_dim * (partial16 ? 2 : 4) + 2 * (size_t)c->cfg.dec_heads * sizeof(float)), 256);
}
extern "C" size_t must3r_hip_cp_slot_bytes(const must3r_hip_ctx* c, int rows) {
    if (!c || rows <= 0) return 0;
    return cp_slot_bytes(c, rows, 0);
}
extern "C" size_t must3r_hip_cp_slot_bytes16(const must3r_hip_ctx* c, int rows) {
    if (!c || rows <= 0) return 0;
    return cp_slot_bytes(c, rows, 1);
}

extern "C" int must3r_hip_decode(must3r_hip_ctx* c, const must3r_hip_decode_args* A, void* stream) {
    if (!c || !A || !A->groups || !A->mem) return fail("decode: null argument");
    if (!c->fin_dec) return fail("decode: decoder weights not finalized");
    const int adt = A->dtype & ~MUST3R_ATTN_FP8;
    if (adt != MUST3R_BF16 && adt != MUST3R_F16 && adt != MUST3R_F16_W2 && adt != MUST3R_F16_WA) return fail("decode: bad dtype %d", A->dtype);
    if ((A->dtype & MUST3R_ATTN_FP8) && !kAttnFp8Built) return fail("decode: MUST3R_ATTN_FP8 is an experiment build (make EXTRA=-DM3R_ATTN_FP8): measured outside the 1e-3 target for +0.4 %%");
    if (A->mem_mode != MUST3R_MEM_KV && A->mem_mode != MUST3R_MEM_NORM_Y && A->mem_mode != MUST3R_MEM_RAW)
        return fail("decode: bad mem_mode %d", A->mem_mode);
    if (A->n_groups <= 0) return fail("decode: no input group");
    if (A->n_mem < 0) return fail("decode: negative n_mem");
    if (A->render && (A->first_call || A->n_mem <= 0)) return fail("decode: render needs a memory (decoder.py:278)");
    if (A->first_call && A->n_mem != 0) return fail("decode: first_call with a non-empty memory");
    if (A->n_scenes < 0) return fail("decode: negative n_scenes");
    const int S = A->n_scenes > 1 ? A->n_scenes : 1;
    if (A->causal != 0 && A->causal != 1) return fail("decode: causal must be 0 or 1");
    if (A->causal && (A->n_groups != 1 || A->cp)) return fail("decode: the causal forward takes ONE group of views (CausalMUSt3R has no list dispatch) and no context parallelism");
    if (A->cp) {   // context-parallel cross attention: the one-view update of the streaming schedule, nothing else
        const must3r_hip_cp& P = *A->cp;
        if (P.world < 1 || P.rank < 0 || P.rank >= P.world) return fail("decode: context parallel: bad rank %d of %d", P.rank, P.world);
        if (A->render || A->first_call || S != 1 || A->n_groups != 1 || A->groups[0].n_views != 1)
            return fail("decode: context parallel is for memory-update calls of ONE view on ONE scene against an existing memory");
        if (A->mem_mode != MUST3R_MEM_KV || (A->dtype & MUST3R_ATTN_FP8)) return fail("decode: context parallel needs memory_mode 'kv' and 16-bit attention operands");
        if (P.n_mem_total <= 0 || A->n_mem > P.n_mem_total) return fail("decode: context parallel: n_mem_total %d with %d local rows", P.n_mem_total, A->n_mem);
        if (P.partial16 != 0 && P.partial16 != 1) return fail("decode: context parallel: partial16 must be 0 or 1");
        if (!P.exchange || !P.slots || (reinterpret_cast<uintptr_t>(P.slots) & 15) || (P.slot_bytes & 15) ||
            P.slot_bytes < cp_slot_bytes(c, A->groups[0].n_tokens, P.partial16))
            return fail("decode: context parallel: slots must be 16-byte aligned, world x slot_bytes with slot_bytes >= %zu, and an exchange function given",
                        cp_slot_bytes(c, A->groups[0].n_tokens, P.partial16));
    }
    long long Rs = 0, views_s = 0;
    for (int gi = 0; gi < A->n_groups; ++gi) {
        const must3r_hip_group& G = A->groups[gi];
        if (!G.tokens || !G.pos || !G.pointmaps) return fail("decode: null buffer in group %d", gi);
        if (G.n_views <= 0 || G.n_tokens <= 0) return fail("decode: empty group %d", gi);
        if (G.pointmaps_scene_stride != 0 && G.pointmaps_scene_stride < (long long)G.n_views * G.H * G.W * 7)
            return fail("decode: group %d: pointmaps_scene_stride %lld is smaller than one scene's pointmaps", gi, (long long)G.pointmaps_scene_stride);
        if (G.pointmaps_scene_stride % 4 != 0 || (reinterpret_cast<uintptr_t>(G.pointmaps) & 15) != 0)   // the head epilogue stores (and accumulates) f32x4 vectors
            return fail("decode: group %d: pointmaps must be 16-byte aligned and pointmaps_scene_stride (%lld) a multiple of 4 floats", gi, (long long)G.pointmaps_scene_stride);
        if (G.H % 16 || G.W % 16 || (G.H / 16) * (G.W / 16) != G.n_tokens)
            return fail("decode: group %d: %dx%d does not give %d tokens", gi, G.H, G.W, G.n_tokens);
        if (G.H / 16 > c->rope_npos || G.W / 16 > c->rope_npos) return fail("decode: group %d: image too large for the RoPE table", gi);
        Rs += (long long)G.n_views * G.n_tokens;
        views_s += G.n_views;
    }
    for (int l = 0; l < c->cfg.dec_depth; ++l)
        if (!A->mem[l]) return fail("decode: null memory buffer for layer %d", l);
    // capacity: rows [0, n_mem) are read, an update writes rows [n_mem, n_mem + Rs) of every scene's buffer
    const long long rows_needed = (long long)A->n_mem + (A->render ? 0 : Rs);
    if (A->mem_capacity <= 0) return fail("decode: mem_capacity must be given (rows each scene's memory buffer can hold)");
    if (rows_needed > A->mem_capacity)
        return fail("decode: memory buffers hold %d rows, this call needs %lld (n_mem %d + %lld new)", A->mem_capacity, rows_needed, A->n_mem,
                    A->render ? 0LL : Rs);
    if (S > 1 && A->mem_scene_stride < A->mem_capacity)
        return fail("decode: mem_scene_stride %lld < mem_capacity %d", (long long)A->mem_scene_stride, A->mem_capacity);
    if ((S - 1) * (S > 1 ? (long long)A->mem_scene_stride : 0) + rows_needed > 0x7fffffffLL || (long long)S * Rs > (1LL << 24))
        return fail("decode: call too large (row indices are 32-bit)");
    // cross-attention staging addresses K|V rows of one scene with 32-bit byte offsets (attention.hip, attn3_kernel)
    {
        const long long memD = A->mem_mode == MUST3R_MEM_KV ? 2LL * c->cfg.dec_dim : 1LL * c->cfg.dec_dim;
        if (rows_needed * 2LL * c->cfg.dec_dim * 2 >= 0x7fffffffLL || rows_needed * memD * 2 >= 0x7fffffffLL)
            return fail("decode: a scene memory of %lld rows exceeds the 2 GiB the attention staging can address per layer", rows_needed);
    }
    // the per-call view tables travel through one staging slot: 2 tables x 24 B per view
    const long long max_views = (long long)(must3r_hip_ctx::kSlotBytes / (2 * sizeof(AttnView)));
    // ... and so does an update's table of K|V destinations: 2 x dec_depth x S pointers ('kv' memory)
    if (!A->render && A->mem_mode == MUST3R_MEM_KV && (size_t)2 * c->cfg.dec_depth * S * sizeof(void*) > must3r_hip_ctx::kSlotBytes)
        return fail("decode: %d scenes in one memory update (limit %zu at decoder depth %d)", S,
                    must3r_hip_ctx::kSlotBytes / (2 * sizeof(void*) * c->cfg.dec_depth), c->cfg.dec_depth);
    if (S * views_s <= max_views) return decode_impl(c, A, stream);
    if (!A->render) return fail("decode: %lld views in one memory update (limit %lld)", S * views_s, max_views);
    if (A->feats) return fail("decode: return_feats with %lld rendered views in one call (limit %lld)", S * views_s, max_views);
    must3r_hip_decode_args sub = *A;
    if (S > 1) {   // cut the batch into ranges of scenes
        const int per = views_s <= max_views ? (int)(max_views / views_s) : 1;
        std::vector<must3r_hip_group> gs(A->n_groups);
        std::vector<void*> mems(c->cfg.dec_depth);
        const long long memRB = ((A->dtype & MUST3R_ATTN_FP8) && A->mem_mode == MUST3R_MEM_KV) ? 3LL * c->cfg.dec_dim
                                : (A->mem_mode == MUST3R_MEM_KV ? 4LL : 2LL) * c->cfg.dec_dim;
        for (int b0 = 0; b0 < S; b0 += per) {
            const int nb = S - b0 < per ? S - b0 : per;
            for (int gi = 0; gi < A->n_groups; ++gi) {
                gs[gi] = A->groups[gi];
                const size_t rg = (size_t)gs[gi].n_views * gs[gi].n_tokens;
                gs[gi].tokens += (size_t)b0 * rg * c->cfg.enc_dim;
                gs[gi].pos += (size_t)b0 * rg * 2;
                gs[gi].pointmaps += (size_t)b0 * (gs[gi].pointmaps_scene_stride > 0 ? (size_t)gs[gi].pointmaps_scene_stride
                                                                                       : (size_t)gs[gi].n_views * gs[gi].H * gs[gi].W * 7);
            }
            for (int l = 0; l < c->cfg.dec_depth; ++l)
                mems[l] = reinterpret_cast<char*>(A->mem[l]) + (size_t)b0 * A->mem_scene_stride * memRB;
            sub.groups = gs.data(); sub.mem = mems.data(); sub.n_scenes = nb;
            M3R_OK(must3r_hip_decode(c, &sub, stream));
        }
        return 0;
    }
    // one scene: cut every group into ranges of views
    for (int gi = 0; gi < A->n_groups; ++gi) {
        const must3r_hip_group& G = A->groups[gi];
        for (long long v0 = 0; v0 < G.n_views; v0 += max_views) {
            must3r_hip_group g1 = G;
            g1.n_views = (int)(G.n_views - v0 < max_views ? G.n_views - v0 : max_views);
            g1.tokens += (size_t)v0 * G.n_tokens * c->cfg.enc_dim;
            g1.pos += (size_t)v0 * G.n_tokens * 2;
            g1.pointmaps += (size_t)v0 * G.H * G.W * 7;
            sub.groups = &g1; sub.n_groups = 1;
            M3R_OK(decode_impl(c, &sub, stream));
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// postprocess + operator-level entry points + profiling
// ------------------------------------------------------------------------------------------------
extern "C" int must3r_hip_postprocess_act(const float* pm, int activation, float* p3, float* pl, float* cf, size_t npix, void* stream) {
    const char* err = "";
    if (!pm || !p3 || !pl || !cf) return fail("postprocess: null argument");
    if (activation != MUST3R_ACT_NORM_EXP && activation != MUST3R_ACT_LINEAR) return fail("postprocess: unknown activation %d", activation);
    if (launch_postprocess(pm, activation == MUST3R_ACT_LINEAR, p3, pl, cf, npix, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}
extern "C" int must3r_hip_postprocess(const float* pm, float* p3, float* pl, float* cf, size_t npix, void* stream) {
    return must3r_hip_postprocess_act(pm, MUST3R_ACT_NORM_EXP, p3, pl, cf, npix, stream);
}

extern "C" int must3r_hip_affine(int is_double, const float* A, const void* sub, const void* B, int b_transposed, const void* bias,
                                 const float* resid, float* out, int M, int N, int K, void* stream) {
    if (M < 0 || N < 0 || K < 0) return fail("affine: negative size");
    if (M == 0 || N == 0) return 0;
    if (!A || !B || !out) return fail("affine: null argument");
    const char* err = nullptr;
    if (launch_gemmx(is_double, A, sub, B, b_transposed, bias, resid, out, M, N, K, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_row_norm(const float* x, int M, int C, float* out, void* stream) {
    if (M < 0 || C <= 0) return fail("row_norm: bad shape");
    if (M == 0) return 0;
    if (!x || !out) return fail("row_norm: null argument");
    const char* err = nullptr;
    if (launch_row_norm(x, M, C, out, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_l2_normalize(const float* x, int64_t outer, int L, int64_t inner, float* out, void* stream) {
    if (outer < 0 || inner < 0 || L < 0) return fail("l2_normalize: negative size");
    if (outer == 0 || inner == 0 || L == 0) return 0;
    if (!x || !out) return fail("l2_normalize: null argument");
    if (outer * inner > (int64_t)0x7fffffff * 4) return fail("l2_normalize: too many vectors");
    const char* err = nullptr;
    if (launch_l2_normalize(x, outer, L, inner, out, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_layernorm_act_f32(const float* x, const float* gamma, const float* beta, float eps, int M, int C, int gelu,
                                            float* out, void* stream) {
    if (M < 0 || C <= 0) return fail("layernorm_act_f32: bad shape");
    if (M == 0) return 0;
    if (!x || !out) return fail("layernorm_act_f32: null argument");
    const char* err = nullptr;
    if (launch_ln_act_f32(x, gamma, beta, eps, M, C, gelu, out, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_topk_gather(const float* feat, const float* attn, int n_images, int N, int C, int k, float* out_feat,
                                      float* out_attn, int64_t* out_idx, void* stream) {
    if (n_images < 0 || N < 0 || C <= 0 || k < 0) return fail("topk_gather: bad shape");
    if (n_images == 0 || k == 0) return 0;
    if (!feat || !attn || !out_feat || !out_attn || !out_idx) return fail("topk_gather: null argument");
    const char* err = nullptr;
    if (launch_topk_gather(feat, attn, n_images, N, C, k, out_feat, out_attn, reinterpret_cast<long long*>(out_idx),
                           reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_weighted_spoc(const float* feat, const float* attn, int n_images, int N, int C, float* out, void* stream) {
    if (n_images < 0 || N < 0 || C <= 0) return fail("weighted_spoc: bad shape");
    if (n_images == 0) return 0;
    if (!feat || !attn || !out) return fail("weighted_spoc: null argument");
    const char* err = nullptr;
    if (launch_weighted_spoc(feat, attn, n_images, N, C, out, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_nn_query(const float* db, int64_t n_db, const float* q, int64_t n_q, float* out, void* stream) {
    if (n_db < 0 || n_q < 0) return fail("nn_query: negative count");
    if (n_q == 0) return 0;
    if (!q || !out || (n_db > 0 && !db)) return fail("nn_query: null argument");
    const char* err = nullptr;
    if (launch_nn_query(db, n_db, q, n_q, out, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_quadrant_ids(const float* pts, int64_t n, const float* cam_center, int divider, int32_t* out, void* stream) {
    if (n < 0) return fail("quadrant_ids: negative count");
    if (n == 0) return 0;
    if (!pts || !cam_center || !out) return fail("quadrant_ids: null argument");
    const char* err = nullptr;
    if (launch_quadrant_ids(pts, n, cam_center, divider, out, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" size_t must3r_hip_postprocess_cam_scratch_bytes(int n_views, int H, int W) {
    if (n_views <= 0 || H <= 0 || W <= 0) return 0;
    return cam_scratch_bytes(n_views, H, W);
}

extern "C" int must3r_hip_postprocess_cam_act(const float* pm, int activation, int n_views, int H, int W, float* p3, float* pl, float* cf,
                                              float* focal, float* c2w, void* scratch, size_t scratch_bytes, void* stream) {
    if (n_views < 0 || H <= 0 || W <= 0) return fail("postprocess_cam: bad shape");
    if (activation != MUST3R_ACT_NORM_EXP && activation != MUST3R_ACT_LINEAR) return fail("postprocess_cam: unknown activation %d", activation);
    if (n_views == 0) return 0;
    if (!pm || !p3 || !pl || !cf || !focal || !c2w || !scratch) return fail("postprocess_cam: null argument");
    const char* err = nullptr;
    if (launch_postprocess_cam(pm, activation == MUST3R_ACT_LINEAR, n_views, H, W, p3, pl, cf, focal, c2w, scratch, scratch_bytes,
                               reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}
extern "C" int must3r_hip_postprocess_cam(const float* pm, int n_views, int H, int W, float* p3, float* pl, float* cf,
                                          float* focal, float* c2w, void* scratch, size_t scratch_bytes, void* stream) {
    return must3r_hip_postprocess_cam_act(pm, MUST3R_ACT_NORM_EXP, n_views, H, W, p3, pl, cf, focal, c2w, scratch, scratch_bytes, stream);
}

extern "C" int must3r_hip_op_gemm(int dtype, int epi, const void* A, const void* W, const float* bias, void* out, int M, int N,
                                  int K, int lda, int ldc, const int64_t* pos, const float* rope_tab, int rope_cols,
                                  int rope_npos, const float* bias2, int row_start2, int accumulate, int ntok, int gw, int H,
                                  int W_img, int wsplit, void* stream) {
    if (dtype != MUST3R_BF16 && dtype != MUST3R_F16) return fail("op_gemm: bad dtype");
    if (epi < 0 || epi >= EPI_COUNT) return fail("op_gemm: bad epilogue");
    GemmArgs a = gargs(A, W, bias, out, M, N, K, lda, ldc);
    a.pos = pos; a.rope_tab = rope_tab; a.rope_cols = rope_cols; a.rope_npos = rope_npos;
    a.bias2 = bias2; a.row_start2 = row_start2; a.accumulate = accumulate;
    a.ntok = ntok; a.gw = gw; a.H = H; a.Wimg = W_img; a.wsplit = wsplit;
    const char* err = "";
    if (launch_gemm((DType)dtype, (Epi)epi, a, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

/* r05, ABI 7: the chip-filling split-weight GEMM with a 2:4-sparse low part (tests / probes): pack, then multiply */
extern "C" int must3r_hip_op_sparse24_pack(const float* w, int rows, int K, void* vals, void* idx, void* stream) {
    const char* err = "";
    if (launch_sparse24_pack(w, rows, K, vals, idx, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}
// r06: the LN fold on the chip-filling 256 x 256 tiles alone (tests): GemmArgs::fold256 with split weights + packed sparse low part (wsplit = 2) or plain fp16 weights
extern "C" int must3r_hip_op_gemm_fold256(int epi, int wsplit, const void* A, const void* W, const void* Wlo_sp, const void* Widx_sp, const float* bias, void* out,
                                          int M, int N, int K, int lda, int ldc, void* x16_out, float* copy32_out, float* stats_out, const float* ln_stats,
                                          const float* ln_s, float ln_eps, float* ln_shift, const int64_t* pos, const float* rope_tab, int rope_cols, int rope_npos,
                                          float out_scale, int scale_cols, void* stream) {
    if (epi < 0 || epi >= EPI_COUNT) return fail("op_gemm_fold256: bad epilogue");
    GemmArgs a = gargs(A, W, bias, out, M, N, K, lda, ldc);
    a.wsplit = wsplit == 2 ? 2 : 0; a.Wlo_sp = Wlo_sp; a.Widx_sp = Widx_sp; a.wsp_rows = N;
    a.x16_out = x16_out; a.copy32_out = copy32_out; a.stats_out = stats_out;
    a.ln_stats = ln_stats; a.ln_s = ln_s; a.ln_eps = ln_eps; a.ln_shift = ln_shift; a.fold256 = 1;
    a.pos = pos; a.rope_tab = rope_tab; a.rope_cols = rope_cols; a.rope_npos = rope_npos;
    a.out_scale = out_scale; a.scale_cols = scale_cols;
    const char* err = "";
    if (launch_gemm(DT_F16, (Epi)epi, a, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}
extern "C" int must3r_hip_op_gemm_sp(int epi, const void* A, const void* W2, const void* Wlo_sp, const void* Widx_sp, const float* bias, void* out, int M,
                                     int N, int K, int lda, int ldc, const int64_t* pos, const float* rope_tab, int rope_cols, int rope_npos, void* stream) {
    if (epi < 0 || epi >= EPI_COUNT) return fail("op_gemm_sp: bad epilogue");
    GemmArgs a = gargs(A, W2, bias, out, M, N, K, lda, ldc);
    a.wsplit = 2; a.Wlo_sp = Wlo_sp; a.Widx_sp = Widx_sp; a.wsp_rows = N;
    a.pos = pos; a.rope_tab = rope_tab; a.rope_cols = rope_cols; a.rope_npos = rope_npos;
    const char* err = "";
    if (launch_gemm(DT_F16, (Epi)epi, a, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_op_gemm_lnfold(int dtype, int epi, const void* A, const void* W2, const float* bias, void* out, int M, int N, int K,
                                         int lda, int ldc, void* x16_out, float* copy32_out, float* stats_out, const float* ln_stats,
                                         const float* ln_s, float ln_eps, float* ln_shift, int ln_shift_init, const int64_t* pos,
                                         const float* rope_tab, int rope_cols, int rope_npos, float out_scale, int scale_cols, void* stream) {
    if (dtype != MUST3R_F16) return fail("op_gemm_lnfold: fp16 operands with split weights only");
    if (epi < 0 || epi >= EPI_COUNT) return fail("op_gemm_lnfold: bad epilogue");
    GemmArgs a = gargs(A, W2, bias, out, M, N, K, lda, ldc);
    a.wsplit = 2;
    a.x16_out = x16_out; a.copy32_out = copy32_out; a.stats_out = stats_out;
    a.ln_stats = ln_stats; a.ln_s = ln_s; a.ln_eps = ln_eps; a.ln_shift = ln_shift; a.ln_shift_init = ln_shift_init;
    a.pos = pos; a.rope_tab = rope_tab; a.rope_cols = rope_cols; a.rope_npos = rope_npos;
    a.out_scale = out_scale; a.scale_cols = scale_cols;
    const char* err = "";
    if (launch_gemm(DT_F16, (Epi)epi, a, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" size_t must3r_hip_attention_scratch_bytes(int nsplit, int total_q_rows, int heads) {
    return attention_split_scratch_bytes(nsplit, total_q_rows, heads);
}

extern "C" int must3r_hip_op_attention(int dtype, const void* Q, const void* K, const void* V, void* O, int ldq, int ldk, int ldv,
                                       int ldo, int heads, const int32_t* views_dev, int n_views, int max_nq, int nsplit,
                                       void* scratch, int total_q_rows, void* stream) {
    const int fp8 = (dtype & MUST3R_ATTN_FP8) ? 1 : 0;   // Q, K, V are e4m3 bytes (strides in bytes); O in the 16-bit type
    if (fp8 && !kAttnFp8Built) return fail("op_attention: MUST3R_ATTN_FP8 is an experiment build (make EXTRA=-DM3R_ATTN_FP8)");
    dtype &= ~MUST3R_ATTN_FP8;
    if (dtype != MUST3R_BF16 && dtype != MUST3R_F16) return fail("op_attention: bad dtype");
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.fp8 = fp8;
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.heads = heads;
    a.views = reinterpret_cast<const AttnView*>(views_dev); a.nviews = n_views; a.max_nq = max_nq; a.scale = 0.125f;
    if (nsplit > 1) {
        if (!scratch || total_q_rows <= 0) return fail("op_attention: split-KV needs scratch and total_q_rows");
        a.nsplit = nsplit; a.total_q_rows = total_q_rows;
        a.part_o = reinterpret_cast<float*>(scratch);
        a.part_ml = a.part_o + (size_t)nsplit * total_q_rows * heads * 64;
    }
    const char* err = "";
    if (launch_attention((DType)dtype, a, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_op_layernorm(int dtype, const float* x, const float* add, const float* w, const float* b, void* out16,
                                       void* out16_lo, float* out32, float* copy32, int M, int C, float eps, void* stream) {
    if (dtype != MUST3R_BF16 && dtype != MUST3R_F16) return fail("op_layernorm: bad dtype");
    LnArgs a = lnargs(x, add, w, b, out16, out16_lo, out32, copy32, M, C, eps);
    const char* err = "";
    if (launch_layernorm((DType)dtype, a, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_op_im2col(int dtype, const float* img, void* out16, int n_views, int H, int W, void* stream) {
    if (dtype != MUST3R_BF16 && dtype != MUST3R_F16) return fail("op_im2col: bad dtype");
    const char* err = "";
    if (launch_im2col((DType)dtype, img, out16, n_views, H, W, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_op_cast(int dtype, const float* in, void* out16, void* out16_lo, size_t n, void* stream) {
    if (dtype != MUST3R_BF16 && dtype != MUST3R_F16) return fail("op_cast: bad dtype");
    const char* err = "";
    if (launch_cast((DType)dtype, in, out16, out16_lo, n, reinterpret_cast<hipStream_t>(stream), &err)) return fail("%s", err);
    return 0;
}

extern "C" int must3r_hip_debug_tr_probe(void* out256_i16_dev, void* stream) {
    if (!out256_i16_dev) return fail("tr_probe: null argument");
    if (launch_tr_probe(reinterpret_cast<short*>(out256_i16_dev), reinterpret_cast<hipStream_t>(stream))) return fail("tr_probe: launch failed");
    return 0;
}

extern "C" int must3r_hip_set_profiling(must3r_hip_ctx* c, int enabled) {
    if (!c) return fail("set_profiling: null context");
    prof_flush(c);
    c->prof = enabled != 0;
    return 0;
}

extern "C" int must3r_hip_get_profile(must3r_hip_ctx* c, must3r_hip_prof_record* out, int max, int reset) {
    if (!c || !out) return 0;
    prof_flush(c);
    int n = 0;
    for (int i = 0; i < PC_COUNT && n < max; ++i) {
        memset(&out[n], 0, sizeof(out[n]));
        strncpy(out[n].name, kProfNames[i], sizeof(out[n].name) - 1);
        out[n].ms = c->prof_ms[i];
        out[n].flops = c->prof_flops[i];
        out[n].calls = c->prof_calls[i];
        ++n;
    }
    // per-kernel rows behind the class rows, names prefixed "k:" (one rocprofv3 symbol each)
    for (auto& kv : c->prof_kern) {
        if (n >= max) break;
        memset(&out[n], 0, sizeof(out[n]));
        snprintf(out[n].name, sizeof(out[n].name), "k:%s", kv.first.c_str());
        out[n].ms = kv.second.ms;
        out[n].flops = kv.second.flops;
        out[n].calls = kv.second.calls;
        ++n;
    }
    if (reset) {
        for (int i = 0; i < PC_COUNT; ++i) { c->prof_ms[i] = 0; c->prof_flops[i] = 0; c->prof_calls[i] = 0; }
        c->prof_kern.clear();
    }
    return n;
}
