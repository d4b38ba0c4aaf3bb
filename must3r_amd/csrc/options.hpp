// Validated A/B switches of libmust3r_hip.  ONE table: name, default, allowed range.  A switch takes its value from
// must3r_hip_set_option (include/must3r_hip.h, ABI 8) or, at first use, from the environment variable M3R_<NAME>; a value outside the range is
// refused (set_option: status 1 + error string; environment: one line on stderr, the default is used).  The switches are measuring instruments:
// every one keeps the results inside the precision mode's tolerance, and all but SPARSE_LO / LNFOLD / LNFOLD256 keep them bit-identical.
#pragma once

namespace m3r {

enum Opt {
    OPT_PERSIST = 0,       // r06: persistent tile loop in the chip-filling GEMM kernels when a launch has more work items than CUs; default 0 = never
                           //      (measured equal to one block per tile: x1.003 over 16 shapes, 526.1 vs 523.6 views/s on the step, profiles/r06_gemm_persist_ab.txt)
    OPT_GEMM256,           // 8-wave 256-row kernels: 0 never, 1 by the fill rule, 2 whenever the shape allows
    OPT_G256K,             // gemm256k_kernel: 0 never, 1 plain-weight launches, 2 split-weight launches too
    OPT_G256P,             // gemm256p_kernel for plain chip-filling launches: 0 never
    OPT_G256P_SPLIT,       // gemm256p_kernel 256 x 128 for dense-low-part split launches: 0 never (also turns the sparse kernels off), 1 where it fills better, 2 always
    OPT_SPARSE_256,        // gemm256s_kernel (256 x 256 sparse tiles) where it fills its rounds: 0 the 256 x 128 sparse form everywhere
    OPT_SPARSE_LO,         // pack and use the 2:4-sparse low part of the split weights (read when a split weight is first packed): 0 dense two-pass kernels everywhere
    OPT_BK128,             // 128-deep K-tiles in gemm48_kernel: 0 the 64-deep form
    OPT_LN_ROWS,           // row-walking LayerNorm for launches of more than 64 k rows: 0 one row per wave everywhere
    OPT_LNFOLD,            // LN fold in one-view update calls: 0 LayerNorm kernels
    OPT_ENC_CHUNK_ROWS,    // token rows per encoder chunk
    OPT_ATTN_LZ,           // attn3_kernel: softmax references move on the tile's row sums (1) or on the per-lane score maxima (0); 2: experiment builds only
    OPT_LNFOLD256,         // r06: LN fold in the chip-filling launches (batched decoder calls, encoder chunks; MUST3R_F16_WA): 0 (default) LayerNorm kernels.
                           //      Measured (profiles/r06_lnfold256_ab.txt, S = 28, interleaved runs on one box): LayerNorm 68.5 -> 13.2 ms per step, GEMMs 534 -> 582 ms
                           //      (producers + 2 B per element of HBM-bound epilogue, consumers + 32 KB of statistics per tile in front of the first DMA): 559 vs 557 views/s
    OPT_G256_GM,           // r06: row-blocks per group of the tile walk of gemm256p / gemm256s (which tiles an XCD's 32 CUs hold together: GM rows x 32 / GM columns)
    OPT_COUNT
};

int opt(Opt o);                                                  // current value (cached after the first call)
int opt_set(const char* name, long long value, const char** err);   // 0 = set; 1 = unknown name / value out of range (*err says which)

}  // namespace m3r
