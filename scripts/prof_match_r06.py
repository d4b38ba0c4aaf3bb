#!/usr/bin/env python
"""Join the evidence of one `scripts/gpu_profile_r06.sh` run into profiles/r06_roofline_evidence.{json,txt}:
   * the bench line of `bench.py --step-only` (per-symbol rows from in-library HIP events: launches, avg us, algorithmic GFLOP, TF/s, frac)
   * the rocprofv3 --kernel-trace --stats summary of the SAME process (per symbol: calls, avg us)
   * the --pmc passes of the same command (per symbol and launch: FETCH_SIZE, WRITE_SIZE, TCC hit rate, MFMA-busy share)
so that every `frac` of the line can be recomputed from files under profiles/ alone:  frac = GFLOP/launch / avg_us(rocprof) / 1e3 / 2500.
Fabric bytes = sum over the L2's memory-side requests by size (TCC_EA0_RDREQ_{32,64,128}B, TCC_EA0_WRREQ_64B / rest 32 B): exact, where the derived
FETCH_SIZE (= RDREQ x 64 B, MI355X_MICROARCH.md "HBM") needs the x2 correction -- and its pass crashes rocprofv3 on this image.  These count MALL hits too."""
import json
import os
import re
import sys

G = "gpurun_out"
FAM = {"g256p": "gemm256p_kernel", "g256ps": "gemm256p_kernel", "g256s": "gemm256s_kernel", "g48k128": "gemm48_kernel", "g256k": "gemm256k_kernel", "g256": "gemm256_kernel", "g256o2": "gemm256_kernel", "g256x": "gemm256x_kernel", "g128": "gemm_kernel", "g64": "gemm_kernel",
       "g64p": "gemm_kernel", "g96": "gemm96_kernel", "g48": "gemm48_kernel", "attn3": "attn3_kernel", "attn4f8": "attn4_kernel"}


def symbol_matches(row_name, sym):
    """does the in-library row name (e.g. g256/e2/w2/n256, attn3/q32/cross) describe the mangled symbol?"""
    parts = row_name.split("/")
    fam = FAM.get(parts[0])
    if not fam or fam not in sym:
        return False
    ints = [int(v) for v in re.findall(r"Li(\d+)E", sym)]
    if parts[0].startswith("attn3"):
        return ints[:1] == [int(parts[1][1:])]
    if parts[0] == "attn4f8":
        return True
    e, w, n = int(parts[1][1:]), int(parts[2][1:]), int(parts[3][1:])
    if parts[0] == "g256s":                              # <T, EPI>: 256 x 256 tiles, sparse low part
        return ints[:1] == [e]
    if parts[0] == "g256ps":                             # gemm256p_kernel<T, EPI, 3, 128, 2, 2>
        return ints[:3] == [e, 3, 128]
    if parts[0] in ("g256p", "g256k", "g256", "g256o2", "g256x"):      # <T, EPI, WS, BN, OCC|ABL|NPH, SYNC>
        ok = ints[:3] == [e, w, n]
        if parts[0] == "g256o2":
            ok = ok and ints[3:4] == [2]
        elif parts[0] == "g256":
            ok = ok and ints[3:4] == [1]
        return ok
    if parts[0] in ("g128", "g64", "g64p"):            # <T, BM, BN, WGM, WGN, EPI, NST, WS, BK, PIPE>
        bm = 128 if parts[0] == "g128" else 64
        return len(ints) >= 9 and ints[0] == bm and ints[1] == n and ints[4] == e and ints[6] == w and ints[8] == (1 if parts[0] == "g64p" else 0)
    if parts[0] == "g96":                               # <T, EPI, NST, PF>
        return ints[:1] == [e]
    if parts[0] in ("g48", "g48k128"):                  # <T, EPI, WS, NST, BK>
        return ints[:2] == [e, w] and ints[3:4] == [128 if parts[0] == "g48k128" else 64]
    return False


def main():
    line = json.loads(open(f"{G}/r06_step_line.json").read().strip().splitlines()[-1])
    stats = {}
    for ln in open(f"{G}/r06_step_kernel_stats.txt"):
        m = re.match(r"(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if m:
            stats[m.group(1)] = {"calls": int(m.group(2)), "total_ms": float(m.group(3)), "avg_us": float(m.group(4)), "pct": float(m.group(5))}
    pmc = {}
    for tag in ("fetch", "write", "dram", "tcc", "sq"):
        try:
            d = json.load(open(f"{G}/r06_pmc_{tag}.json"))
            pmc[tag] = d
        except Exception:
            pmc[tag] = None
    rows = []
    for src in ("roofline", "roofline_attention"):
        for r in (line.get(src) or {}).get("per_symbol", []):
            syms = [s for s in stats if symbol_matches(r["kernel"], s)]
            row = dict(r)
            row["symbols"] = syms
            if syms:
                calls = sum(stats[s]["calls"] for s in syms)
                tot = sum(stats[s]["total_ms"] for s in syms)
                row["rocprof_calls_whole_process"] = calls
                row["rocprof_avg_us"] = round(tot * 1e3 / max(1, calls), 2)
                row["frac_from_rocprof"] = round(r["algorithmic_gflop_per_launch"] / row["rocprof_avg_us"] * 1e3 / 2500.0, 4)   # GF / us = 1000 TF/s
            for tag, d in pmc.items():
                if not d:
                    continue
                per = [v for k, v in d["per_launch_means"].items() if any(symbol_matches(r["kernel"], k) for _ in [0])]
                if not per:
                    continue
                n = sum(p["launches"] for p in per)
                mean = lambda c: sum(p.get(c, 0.0) * p["launches"] for p in per) / max(1, n)  # noqa: E731
                if tag == "fetch":
                    # L2 -> fabric read requests by size (exact bytes: no halving correction needed, unlike the derived FETCH_SIZE = RDREQ x 64 B)
                    r32, r64, r128, rall = mean("TCC_EA0_RDREQ_32B_sum"), mean("TCC_EA0_RDREQ_64B_sum"), mean("TCC_EA0_RDREQ_128B_sum"), mean("TCC_EA0_RDREQ_sum")
                    row["TCC_EA0_RDREQ_per_launch"] = {"all": int(rall), "32B": int(r32), "64B": int(r64), "128B": int(r128)}
                    row["fabric_read_bytes_per_launch"] = int(32 * r32 + 64 * r64 + 128 * r128 + 64 * max(0.0, rall - r32 - r64 - r128))
                elif tag == "write":
                    w64, wall = mean("TCC_EA0_WRREQ_64B_sum"), mean("TCC_EA0_WRREQ_sum")
                    row["TCC_EA0_WRREQ_per_launch"] = {"all": int(wall), "64B": int(w64)}
                    row["fabric_write_bytes_per_launch"] = int(64 * w64 + 32 * max(0.0, wall - w64))
                elif tag == "dram":
                    row["RDREQ_DRAM_share"] = round(mean("TCC_EA0_RDREQ_DRAM_sum") / max(1.0, mean("TCC_EA0_RDREQ_sum")), 4)
                elif tag == "tcc":
                    h, m_ = mean("TCC_HIT_sum"), mean("TCC_MISS_sum")
                    row["TCC_hit_rate"] = round(h / max(1.0, h + m_), 4)
                    row["TCC_requests_per_launch"] = int(h + m_)
                elif tag == "sq":
                    busy, act = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("GRBM_GUI_ACTIVE")
                    row["SQ_VALU_MFMA_BUSY_CYCLES"] = int(busy)
                    row["GRBM_GUI_ACTIVE"] = int(act)
                    # SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (= 16 x SQ_INSTS_MFMA for the 16x16x32 MFMA), GRBM_GUI_ACTIVE over the
                    # 8 XCDs (cross-check: SQ_BUSY_CYCLES / 32 shader engines gives the same kernel cycles): share of time a SIMD's matrix pipe is busy
                    row["kernel_cycles"] = int(act / 8.0)
                    row["mfma_pipe_busy_share"] = round(busy / 1024.0 / max(1.0, act / 8.0), 4)
                    row["valu_per_mfma"] = round(mean("SQ_INSTS_VALU") / max(1.0, mean("SQ_INSTS_MFMA")), 2)
                row.setdefault("pmc_scenes", d["scenes"])
            if "fabric_read_bytes_per_launch" in row and "fabric_write_bytes_per_launch" in row:
                row["fabric_bytes_per_launch_corrected"] = row["fabric_read_bytes_per_launch"] + row["fabric_write_bytes_per_launch"]
            rows.append(row)
    # rows that share a rocprofv3 symbol (self and cross attention run the same kernel): the trace cannot tell them apart, so the recomputed
    # fraction is that of the symbol as a whole -- sum of the rows' algorithmic GFLOP over the symbol's total time
    groups = {}
    for r in rows:
        if r.get("symbols"):
            groups.setdefault(tuple(r["symbols"]), []).append(r)
    for syms, rs in groups.items():
        if len(rs) > 1:
            gf = sum(r["algorithmic_gflop_per_launch"] * r["launches"] for r in rs)
            n = sum(r["launches"] for r in rs)
            us = rs[0]["rocprof_avg_us"]
            for r in rs:
                r["frac_from_rocprof"] = round(gf / n / us * 1e3 / 2500.0, 4)
                r["frac_from_rocprof_note"] = "symbol shared by " + " + ".join(x["kernel"] for x in rs) + ": fraction of the symbol as a whole"
    commit = os.environ.get("M3R_COMMIT", "?")
    doc = {"commit": commit, "pmc_pass_commits": {t: (d or {}).get("commit") for t, d in pmc.items()}, "command_trace": "rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 3 --warmup 1 --scenes S --step-only",
           "command_pmc": "rocprofv3 --kernel-trace --pmc <set> -- python bench.py --gpus 1 --steps 1 --warmup 1 --scenes S_pmc --step-only (one pass per set)",
           "line": {k: line.get(k) for k in ("value", "ms_per_step", "config", "roofline", "roofline_attention", "kernel_classes", "end_to_end_mfma_frac")},
           "rows": rows,
           "how_to_recompute": "frac = algorithmic_gflop_per_launch / avg_us * 1000 / 2500 (1 GF per us = 1000 TF/s; peak 2500 TF/s) with avg_us from r06_step_kernel_stats.txt (rocprof_avg_us; the trace also "
                               "holds the warm-up / stage-split passes of the same step, same launch mix); fabric bytes = 32/64/128-byte read requests + 64/32-byte write requests of the L2 (PMC passes "
                               "at pmc_scenes scenes in flight: per-launch figures scale with the rows per launch, compare like with like)"}
    os.makedirs("profiles", exist_ok=True)
    json.dump(doc, open(f"{G}/r06_roofline_evidence.json", "w"), indent=1)
    with open(f"{G}/r06_roofline_evidence.txt", "w") as f:
        f.write(f"# commit {commit}; value {line.get('value')} views/s; per-symbol rows of the step (bench line = in-library HIP events; rocprof = kernel trace of the same process)\n")
        f.write(f"{'kernel':22s} {'launches':>8s} {'avg_us':>9s} {'rocprof':>9s} {'GF/launch':>10s} {'TF/s':>7s} {'frac':>6s} {'frac_rp':>7s} {'MB/launch':>10s} {'TCC hit':>8s} {'MFMA busy':>9s} {'VALU/MFMA':>9s}\n")
        for r in rows:
            f.write(f"{r['kernel']:22s} {r['launches']:8d} {r['avg_launch_us']:9.2f} {r.get('rocprof_avg_us', float('nan')):9.2f} {r['algorithmic_gflop_per_launch']:10.2f} "
                    f"{r['achieved_tflops']:7.1f} {r['frac']:6.3f} {r.get('frac_from_rocprof', float('nan')):7.3f} {r.get('fabric_bytes_per_launch_corrected', 0) / 1e6:10.1f} "
                    f"{r.get('TCC_hit_rate', float('nan')):8.3f} {r.get('mfma_pipe_busy_share', float('nan')):9.3f} {r.get('valu_per_mfma', float('nan')):9.2f}\n")
    print(open(f"{G}/r06_roofline_evidence.txt").read())


if __name__ == "__main__":
    main()
