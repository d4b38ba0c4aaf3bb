#!/usr/bin/env python
"""Headline benchmark: views/sec for MUSt3R_512 (ViT-L encoder / ViT-B memory decoder), 20-view 512x384 scenes.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM: S (``--scenes``, default 28; 20 in r03-r04)
independent scenes of 20 views per rank -- every scene: encode 20, memory update with the demo schedule [2,1,...,1] (20-view
memory), render 20 against its final memory, fp32 activation (BASELINE.json configs[2]).  The S scenes
are IN FLIGHT TOGETHER: they ride the batch dimension of the reference's decoder API (decoder.py:170-186), one native call per
schedule step, so the strictly sequential memory update runs its GEMMs on S x 768 rows instead of 768.
  N = 1 : ``value`` = S x 20 views per step / time.  ``single_scene`` carries the S = 1 numbers (one scene at a time: the
          figure of rounds 1-2) with its stage split and kernel classes.
  N > 1 : REPLICAS -- every rank runs its own S scenes (the metric's whole-node definition, SURVEY.md section 8e: independent
          scenes shard with no data-path collective); barrier + max over ranks around the timed region.  The view-SHARDED
          forms of one scene (RCCL all-gather of the encoded keyframe tokens over xGMI, replicated sequential update:
          must3r_amd/parallel.py) are timed in the same run and reported under ``configs``.
Prints ONE JSON line on rank 0.  ``configs`` carries the other BASELINE.json configurations measured in the same run (never
part of ``value``):
  configs[1]  MUSt3R_224 10-view 224x224 scene
  configs[3]  200-frame online streaming memory (N = 1: one GPU; N > 1: frames sharded, all-gather, replicated update)
  configs[4]  mixed-resolution scene 512 x {384,336,288,256,160} x 4 views through forward_list, 16-bit and fp8 (MX) attention
  N > 1 only: the weak- and strong-scaling view-sharded forms of the 20-view scene.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp16w2": 2500.0, "fp16wa": 2500.0}   # dense MFMA peak, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
SUSTAINED_TFLOPS = 1750.0   # measured: MFMA-only loop, random fp16 operands, 10 ms (profiles/r03_pipe_rates.txt)
MIXED_H = (384, 336, 288, 256, 160)                                 # BASELINE.json configs[4], W = 512, 4 views each


def build_models(cfg, precision, device):
    import must3r_amd.model as M
    from must3r_amd import synthetic as S
    enc = M.Dust3rEncoder(img_size=(cfg.img_size,) * 2, embed_dim=cfg.enc_dim, depth=cfg.enc_depth, num_heads=cfg.enc_heads,
                          precision=precision)
    dec = M.MUSt3R(img_size=(cfg.img_size,) * 2, enc_embed_dim=cfg.enc_dim, embed_dim=cfg.dec_dim, depth=cfg.dec_depth,
                   num_heads=cfg.dec_heads, feedback_type="single_mlp", memory_mode="kv", landscape_only=False,
                   precision=precision)
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    enc.load_state_dict(sde, strict=True)
    dec.load_state_dict(sdd, strict=True)
    return enc.to(device).eval(), dec.to(device).eval(), sde, sdd


def view_flops(N):
    """Algorithmic FLOPs per view: encoder, decoder fixed part, cross attention per attended
    memory token, memory write."""
    enc = 2 * N * 768 * 1024 + 24 * (24 * N * 1024 ** 2 + 4 * N * N * 1024)
    dec_fixed = 2 * N * 1024 * 768 + 12 * (28 * N * 768 ** 2 + 4 * N * N * 768) + 2 * N * 768 * 1792
    ca = 12 * 4 * N * 768
    memw = 12 * 4 * N * 768 ** 2 + 16 * N * 768 ** 2
    return enc, dec_fixed, ca, memw


def scene_flops(N, V, K):
    """One scene: V views encoded + rendered, K-keyframe memory built with [2,1,...,1]."""
    enc, dec_fixed, ca, memw = view_flops(N)
    upd_ca = ca * N * (2 + sum(range(2, K)))   # init: 2 views x 1 other view; then view i attends i previous views
    return V * enc + K * (dec_fixed + memw) + upd_ca + 2 * 12 * 4 * N * 768 ** 2 + V * dec_fixed + V * ca * N * K


def mixed_scene_flops(tokens):
    """Mixed-resolution scene: view i has tokens[i] tokens; memory = all views, schedule [2,1,...,1], render all."""
    total, mem_tok = 0, 0
    for i, N in enumerate(tokens):
        enc, dec_fixed, ca, memw = view_flops(N)
        total += enc + dec_fixed + memw
        if i == 1:
            total += ca * tokens[0] + view_flops(tokens[0])[2] * N + 12 * 4 * (tokens[0] + N) * 768 ** 2   # init pair attends each other (pre-feedback K|V)
        elif i > 1:
            total += ca * mem_tok
        mem_tok += N
    for N in tokens:
        enc, dec_fixed, ca, memw = view_flops(N)
        total += dec_fixed + ca * mem_tok
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--views", type=int, default=20)
    ap.add_argument("--scenes", type=int, default=28, help="S: independent 20-view scenes in flight together per rank (they ride the "
                    "decoder's batch dimension: M = S x 768 rows in the sequential memory update); 1 = one scene at a time.  r05 default 28 (r03-r04: "
                    "20): at 28 every decoder GEMM of the update fills its rounds of 256-row tiles over the 256 CUs to 98 %% (84 row blocks x 9 / 12 / 3 / 6 "
                    "column tiles; at 20: 70-94 %%) -- DESIGN.md section 3.4, profiles/r04_scenes_sweep.txt; the S = 20 figure stays in "
                    "scenes_in_flight_sweep for continuity with BENCH_r03 / r04")
    ap.add_argument("--precision", default="fp16wa", choices=["bf16", "fp16", "fp16w2", "fp16wa"],
                    help="MFMA operand mode; fp16wa (fp16, split weights except in the Mlp Linears) and fp16w2 (all weights split) meet the "
                         "1e-3 parity target")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=20, help="views of the scene the CPU oracle runs (20 = the metric's own workload)")
    ap.add_argument("--no-alt", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the one-scene-at-a-time leg (the PMC passes use this)")
    ap.add_argument("--no-configs", action="store_true", help="skip the extra BASELINE.json configurations")
    ap.add_argument("--step-only", action="store_true", help="the S-scene step and nothing else (no single-scene leg, other configurations, other "
                    "precisions, cam timing or CPU baseline): every launch of the process belongs to the step, so a rocprofv3 --kernel-trace of "
                    "this command has per-symbol averages that equal the line's (scripts/gpu_profile_r04.sh)")
    ap.add_argument("--stream-frames", type=int, default=200)
    ap.add_argument("--cpu-timeout", type=float, default=420.0)
    args = ap.parse_args()
    if args.step_only:
        args.no_single = args.no_configs = args.no_alt = args.no_cpu_baseline = True

    import torch.distributed as dist
    from must3r_amd.config import MUST3R_512, MUST3R_224
    from must3r_amd import synthetic as S
    from must3r_amd.engine import run_scene, run_scenes, run_scene_mixed, run_scenes_mixed, run_video, demo_mem_batches
    from must3r_amd.parallel import run_scene_sharded, run_video_sharded, shard_range

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # host threads: the CPU only generates the synthetic weights; never let N ranks x all cores oversubscribe the box
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // max(1, world))))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("M3R_DIST_BACKEND", "nccl")   # "gloo": ranks may share a GPU (1-GPU box dry run of the N>1 path)
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # the N > 1 line must say which transport carried its collectives: RCCL ("nccl" on ROCm) unless M3R_DIST_BACKEND asked for the dry-run one
        assert dist.get_backend() == backend and (backend == "nccl" or "M3R_DIST_BACKEND" in os.environ), (dist.get_backend(), backend)

    cfg, H, W, V = MUST3R_512, 384, 512, args.views
    N = (H // 16) * (W // 16)
    Sn = max(1, args.scenes)
    enc, dec, sde, sdd = build_models(cfg, args.precision, device)
    # scene 0 of rank 0 is the seed-0 scene the CPU oracle runs; every other (rank, scene) has its own seed.  All resident in HBM.
    ts = S.make_images(V, H, W, seed=0)[1]
    scenes = torch.stack([S.make_images(V, H, W, seed=(0 if (rank == 0 and b == 0) else 1000 + rank * 64 + b))[0] for b in range(Sn)]).to(device)
    imgs = scenes[0]
    tdt = torch.bfloat16 if args.precision == "bf16" else torch.float16
    dtype_label = {"bf16": "bf16", "fp16": "fp16", "fp16w2": "fp16 (split weights)",
                   "fp16wa": "fp16 (split weights in the attention-side Linears, plain in the Mlp Linears)"}[args.precision]

    def step():
        return run_scenes(enc, dec, scenes, ts) if Sn > 1 else run_scene(enc, dec, imgs, ts)

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    def timed(fn, nsteps):
        sync()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            fn()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(max(args.warmup, 1)):
        step()
    dt = timed(step, args.steps)
    views_per_step = Sn * V * world
    value = views_per_step * args.steps / dt

    def profile_classes(fn):
        """per-kernel-class timing with HIP events on the launch stream (one extra, untimed pass of fn)"""
        for m in (enc, dec):
            m._context().set_profiling(True)
        fn()
        torch.cuda.synchronize(device)
        prof, kern_rows = {}, {}
        for m in (enc, dec):
            for k, v in m._context().get_profile().items():
                p = (kern_rows if k.startswith("k:") else prof).setdefault(k[2:] if k.startswith("k:") else k, {"ms": 0.0, "flops": 0.0, "calls": 0})
                p["ms"] += v["ms"]; p["flops"] += v["flops"]; p["calls"] += v["calls"]
            m._context().set_profiling(False)
        classes = {k: {"ms": round(v["ms"], 3), "calls": int(v["calls"]),
                       "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 and v["flops"] > 0 else None}
                   for k, v in prof.items()}
        prof["_kernels"] = kern_rows
        return prof, classes

    def stage_split(n_scenes, warm=True):
        if warm:   # the pass allocates its own memory buffers (mem = None): a first pass may pay hipMalloc for them, the second finds them cached
            stage_split(n_scenes, warm=False)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ts_host = ts.cpu()
        ev[0].record()
        x, pos = enc(scenes[:n_scenes].reshape(n_scenes * V, 3, H, W), ts_host.repeat(n_scenes, 1))
        ev[1].record()
        x, pos = x.view(n_scenes, V, *x.shape[1:]), pos.view(n_scenes, V, *pos.shape[1:])
        tsb = ts_host.unsqueeze(0).expand(n_scenes, -1, -1)
        mem, i = None, 0
        for nb in demo_mem_batches(V):
            mem, _ = dec(x[:, i:i + nb], pos[:, i:i + nb], tsb[:, i:i + nb], mem)
            i += nb
        ev[2].record()
        dec(x, pos, tsb, mem, render=True)
        ev[3].record()
        torch.cuda.synchronize(device)
        return {"encode": round(ev[0].elapsed_time(ev[1]), 2), "update": round(ev[1].elapsed_time(ev[2]), 2),
                "render": round(ev[2].elapsed_time(ev[3]), 2)}

    # ---- per-kernel-class timing of the step (one extra, untimed step) and its stage split
    prof, classes = profile_classes(step)
    stages = stage_split(Sn) if world == 1 else None

    # ---- one scene at a time (S = 1): the figure of rounds 1-2, kept beside the headline
    single = None
    if world == 1 and Sn > 1 and not args.no_single:
        fn1 = lambda: run_scene(enc, dec, imgs, ts)  # noqa: E731
        fn1()
        d1 = timed(fn1, args.steps)
        _, cls1 = profile_classes(fn1)
        single = {"value": round(V * args.steps / d1, 2), "unit": "views/s", "ms_per_step": round(d1 / args.steps * 1e3, 3),
                  "workload": f"ONE {V}-view 384x512 scene at a time (encode {V}, update [2,1,...,1] one view per call = M 768 GEMMs, render {V})",
                  "stages_ms": stage_split(1), "kernel_classes": cls1,
                  "end_to_end_mfma_frac": round(scene_flops(N, V, V) * args.steps / d1 / 1e12 / PEAK_TFLOPS[args.precision], 4)}

    # roofline of the dominant KERNEL = one symbol of the rocprofv3 trace.  attn3_kernel (self + cross launches) is the top
    # symbol in every precision; the GEMM template is spread over one symbol per epilogue, so its two tile classes are
    # reported next to it under "roofline_gemm".
    kern = {"attn3_kernel": (["attn_self", "attn_cross"], ("attn3_kernel", "attn4_kernel")),
            "gemm_kernel<big tile>": (["gemm128"], ("gemm256_kernel", "gemm256k_kernel", "Li128ELi64E", "Li128ELi128E")),
            "gemm_kernel<small-M>": (["gemm64"], ("Li64ELi64E", "gemm48_kernel", "gemm96_kernel", "gemms_kernel"))}
    # (r05, VERDICT r04 item 3: `traffic` comes from THIS round's PMC passes at THIS run's number of scenes in flight or is null -- see the evidence
    # block below; the r01-r03 per-class file profiles/rNN_pmc_traffic.json is no longer read)
    pmc_file, pmc, pmc_doc = None, {}, {}
    want = "DF16b" if args.precision == "bf16" else "DF16_"

    def roof(name):
        cs, sym = kern[name]
        a = {f: sum(prof[c][f] for c in cs if c in prof) for f in ("ms", "flops", "calls")}
        ach = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0.0
        r = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
             "frac": round(ach / PEAK_TFLOPS[args.precision], 4), "traffic": None,
             "avg_launch_us": round(a["ms"] * 1e3 / max(1, a["calls"]), 2), "launches": int(a["calls"]),
             "algorithmic_flops_per_launch": round(a["flops"] / max(1, a["calls"]) / 1e9, 3), "flops_unit": "GFLOP",
             # what the matrix pipe sustains on THIS chip with random 16-bit operands in an MFMA-only loop (power-limited clock;
             # all-ones operands: 2190-2260): profiles/r03_pipe_rates.txt, scripts/probes/pipe_rates.hip
             "sustained_mfma_only": SUSTAINED_TFLOPS, "frac_of_sustained": round(ach / SUSTAINED_TFLOPS, 4)}
        # HBM bytes per launch: FETCH_SIZE / WRITE_SIZE cannot be read from inside the process; they come from the committed
        # PMC passes of this same command (scripts/gpu_pmc.sh + scripts/pmc_summary.py -> profiles/rNN_pmc_traffic.json)
        rows = [v for k, v in pmc.items() if any(t in k for t in sym) and want in k]
        if rows and pmc_doc.get("scenes", 8) != Sn:   # per-launch bytes of another step size do not belong to these launches
            r["traffic_note"] = f"{os.path.relpath(pmc_file, ROOT)} was taken at {pmc_doc.get('scenes', 8)} scenes in flight, this run has {Sn}"
        elif rows:
            n = sum(x["launches"] for x in rows)
            r["traffic"] = int(sum(x["hbm_bytes_per_launch_corrected"] * x["launches"] for x in rows) / max(1, n))
            r["traffic_unit"] = "bytes/launch (PMC, corrected)"
            r["traffic_source"] = (os.path.relpath(pmc_file, ROOT) + " (separate --pmc passes of this command, taken at commit "
                                   + str(pmc_doc.get("commit", "?")) + "; not this run)")
        return r

    def symbol_rows(prefixes):
        """one row per kernel symbol of the step (in-library HIP events around every launch): what a rocprofv3 --kernel-trace --stats of
        `bench.py --step-only` lists under the same symbol (profiles/r04_step_kernel_stats.txt; scripts/prof_match.py joins the two)"""
        rows = []
        for name, v in sorted(prof.get("_kernels", {}).items(), key=lambda kv: -kv[1]["ms"]):
            if not name.startswith(prefixes) or v["ms"] <= 0:
                continue
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
            rows.append({"kernel": name, "launches": int(v["calls"]), "ms": round(v["ms"], 3), "avg_launch_us": round(v["ms"] * 1e3 / max(1, v["calls"]), 2),
                         "algorithmic_gflop_per_launch": round(v["flops"] / max(1, v["calls"]) / 1e9, 3), "achieved_tflops": round(tf, 1),
                         "frac": round(tf / PEAK_TFLOPS[args.precision], 4)})
        return rows

    roofline_attention = roof("attn3_kernel")
    roofline_attention["per_symbol"] = symbol_rows(("attn",))
    roofline_gemm = [r for r in (roof("gemm_kernel<big tile>"), roof("gemm_kernel<small-M>")) if r["launches"] > 0]   # (no all-zero row for a class the step never launches)
    # `roofline` = the kernel CLASS that takes the most time of the step (the GEMM family, not the single top symbol)
    gemm_ms = sum(prof[c]["ms"] for c in ("gemm128", "gemm64") if c in prof)
    attn_ms = sum(prof[c]["ms"] for c in ("attn_self", "attn_cross") if c in prof)
    if gemm_ms >= attn_ms:
        a = {f: sum(prof[c][f] for c in ("gemm128", "gemm64") if c in prof) for f in ("ms", "flops", "calls")}
        ach = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0.0
        roofline = {"bound": "mfma", "kernel": "GEMM family (every out = epi(A W^T + b) launch of the step: gemm256p / gemm256s / gemm256k / gemm256 / gemm_kernel / gemm96 / gemm48 symbols)",
                    "achieved": round(ach, 1), "peak": PEAK_TFLOPS[args.precision], "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS[args.precision], 4),
                    "traffic": None,
                    "ms_per_step": round(a["ms"], 3), "share_of_step_kernel_time": round(a["ms"] / max(1e-9, sum(v["ms"] for k, v in prof.items() if k != "_kernels")), 4),
                    "launches": int(a["calls"]), "avg_launch_us": round(a["ms"] * 1e3 / max(1, a["calls"]), 2),
                    "algorithmic_flops": "2 M N K per launch, ONE pass (split-weight launches run two MFMA passes for it)",
                    "sustained_mfma_only": SUSTAINED_TFLOPS, "frac_of_sustained": round(ach / SUSTAINED_TFLOPS, 4),
                    "per_symbol": symbol_rows(("g",))}
    else:
        roofline = roofline_attention
    # fabric bytes per launch (L2 -> MALL / HBM requests by size, PMC) from the closing profile of the round, joined per symbol by
    # scripts/prof_match.py -- only when it was taken at this step size (per-launch bytes scale with the rows per launch)
    EV = next((f for f in ("profiles/r06_roofline_evidence.json", "profiles/r05_roofline_evidence.json") if os.path.exists(os.path.join(ROOT, f))),
              "profiles/r06_roofline_evidence.json")
    try:
        ev = json.load(open(os.path.join(ROOT, EV)))
        # the passes must belong to this code: their commit has to be an ancestor of HEAD wherever a git checkout is there to ask (the GPU box runs a
        # snapshot without .git: there the file's own commit stamp is what the line quotes)
        anc = None
        try:
            import subprocess
            if os.path.isdir(os.path.join(ROOT, ".git")) and ev.get("commit") not in (None, "?"):
                anc = subprocess.run(["git", "-C", ROOT, "merge-base", "--is-ancestor", str(ev["commit"]), "HEAD"], capture_output=True, timeout=20).returncode == 0
        except Exception:
            anc = None
        by = {r["kernel"]: r for r in ev["rows"] if r.get("fabric_bytes_per_launch_corrected")} if anc is not False else {}
        for rf in (roofline, roofline_attention):
            tot_b = tot_n = 0
            for row in rf.get("per_symbol", []):
                e = by.get(row["kernel"])
                if e and e.get("pmc_scenes") == Sn:
                    row["traffic"] = e["fabric_bytes_per_launch_corrected"]
                    row["TCC_hit_rate"], row["mfma_pipe_busy_share"] = e.get("TCC_hit_rate"), e.get("mfma_pipe_busy_share")
                    tot_b += row["traffic"] * row["launches"]; tot_n += row["launches"]
            if tot_n:
                rf["traffic"] = int(tot_b / tot_n)
                rf["traffic_unit"] = "fabric bytes per launch (sized TCC_EA0 read / write requests, launch-weighted over the class's symbols)"
                rf["traffic_source"] = (f"{EV} (separate --pmc passes of `bench.py --step-only --scenes {Sn}`, commit {ev.get('commit', '?')}"
                                        + (", an ancestor of this checkout's HEAD" if anc else "") + "; not this run)")
            else:
                rf["traffic"] = None
                rf["traffic_note"] = (f"{EV} is not an ancestor of HEAD" if anc is False else
                                      f"{EV} holds no pass at this run's {Sn} scenes in flight" if by else f"{EV}: no usable rows")
    except Exception:
        for rf in (roofline, roofline_attention):
            rf["traffic"] = None
            rf["traffic_note"] = f"{EV} not found: no PMC pass of this round to quote"

    # postprocess(compute_cam=True) on one scene's 20 rendered pointmaps (HBM-bound:
    # 28 B read + 28 B written per pixel; the focal iteration and the registration add no HBM pass)
    cam = None
    if rank == 0 and world == 1 and not args.step_only:
        from must3r_amd.engine import postprocess
        pmaps = run_scene(enc, dec, imgs, ts)["render"]
        for _ in range(3):
            postprocess(pmaps, compute_cam=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            postprocess(pmaps, compute_cam=True)
        e1.record()
        torch.cuda.synchronize(device)
        ms = e0.elapsed_time(e1) / reps
        nbytes = pmaps.numel() // 7 * 56
        cam = {"op": "postprocess(compute_cam=True): activation + Weiszfeld focal + weighted rigid registration",
               "views": V, "ms": round(ms, 4), "bound": "hbm", "achieved": round(nbytes / ms / 1e6, 1), "peak": 8000.0,
               "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / 8000.0, 4), "algorithmic_bytes": nbytes}
        del pmaps

    # the step at another number of scenes in flight: 28 fills the rounds of every decoder GEMM of the update to 98 % (84 row blocks x 9 / 12 / 3 / 6
    # column tiles over 256 CUs; at 20: 70-94 %) -- reported beside `value`, never in it (DESIGN.md section 3.4 "round quantisation")
    sweep = None
    if not args.no_alt and world == 1 and Sn in (20, 28):
        S2 = 28 if Sn == 20 else 20   # the other of the two reported step sizes (r03-r04 headline: 20; r05: 28)
        if S2 > Sn:
            more = torch.stack([S.make_images(V, H, W, seed=2000 + b)[0] for b in range(S2 - Sn)]).to(device)
            scenes2 = torch.cat([scenes, more], dim=0)
            del more
        else:
            scenes2 = scenes[:S2]
        fn2 = lambda: run_scenes(enc, dec, scenes2, ts)  # noqa: E731
        fn2()
        k2 = max(2, args.steps // 2)
        d2 = timed(fn2, k2)
        sweep = [{"scenes_in_flight": S2, "value": round(S2 * V * k2 / d2, 2), "ms_per_step": round(d2 / k2 * 1e3, 3),
                  "note": "the r03-r04 definition of `value`" if S2 == 20 else "the r05 definition of `value`"}]
        del scenes2

    alt = None
    if not args.no_alt and world == 1:
        alt = []
        for other in ("bf16", "fp16", "fp16w2", "fp16wa"):
            if other == args.precision:
                continue
            enc.precision = dec.precision = other
            step()
            dta = timed(step, max(2, args.steps // 2))
            alt.append({"dtype": other, "value": round(views_per_step * max(2, args.steps // 2) / dta, 2)})
        enc.precision = dec.precision = args.precision

    # ---- the other BASELINE.json configurations, measured in the same run (reported beside the headline, never in it)
    configs = []
    rccl, sharded = None, None   # N > 1: what the collectives carried and the view-sharded figures, at the TOP level of the line next to the replica `value`
    if not args.no_configs:
        ksteps = max(2, min(10, args.steps))
        if world == 1:
            # configs[4]: mixed resolution through forward_list, 16-bit attention and the fp8 (MX-scaled Q K^T) attention path
            groups = [S.make_images(4, h, 512, seed=100 + gi)[0].to(device) for gi, h in enumerate(MIXED_H)]
            tokens = [(h // 16) * 32 for h in MIXED_H for _ in range(4)]
            fl = mixed_scene_flops(tokens)
            mixed = {"config": "configs[4] MUSt3R_512 mixed-resolution scene 512x{384,336,288,256,160} x4 views (forward_list), one scene at a time",
                     "views_per_step": 20, "unit": "views/s", "scene_tflop": round(fl / 1e12, 2), "modes": []}
            ref_mixed = None
            from must3r_amd import _lib as _m3r_lib
            fp8_built = _m3r_lib.has_fp8_attention()   # r06: the e4m3 attention path is parked (include/must3r_hip.h MUST3R_ATTN_FP8): timed only on an experiment build
            if not fp8_built:
                mixed["fp8_attention"] = ("parked in r06: 680.4 vs 677.9 views/s (+0.4 %) at 1.19e-3 / 1.40e-3 from the 16-bit path with 8 scenes in flight (BENCH_r05) -- "
                                          "outside the 1e-3 target; built only with make EXTRA=-DM3R_ATTN_FP8")
            for prec, fp8 in ((args.precision, False), (args.precision, True)) if fp8_built else ((args.precision, False),):
                enc.precision = dec.precision = prec
                enc.attention_fp8 = dec.attention_fp8 = fp8
                fnm = lambda: run_scene_mixed(enc, dec, groups)  # noqa: E731
                om = fnm()
                d = timed(fnm, ksteps)
                mode = {"dtype": prec + (" + fp8 attention (e4m3 Q / K through the MX-scaled 32x32x64 MFMA; [K e4m3 | V fp16] memory rows)" if fp8 else ""),
                        "value": round(20 * ksteps / d, 2), "ms_per_step": round(d / ksteps * 1e3, 3), "mfma_frac": round(fl * ksteps / d / 1e12 / 2500.0, 4)}
                if not fp8:
                    ref_mixed = [r.clone() for r in om["render"]]
                else:   # the mode's distance from the 16-bit path on the same scene (the 16-bit path's own error vs the oracle: parity_vs_cpu_oracle)
                    mode["render_rel_inf_vs_16bit_path"] = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(om["render"], ref_mixed))
                mixed["modes"].append(mode)
            enc.precision = dec.precision = args.precision
            enc.attention_fp8 = dec.attention_fp8 = False
            del ref_mixed, om
            # ... and with Sm mixed-resolution scenes IN FLIGHT (forward_list with B = Sm): every attention launch is chip-filling, the only
            # regime in which the 32 x 32 fp8 kernel can beat the 16-bit one -- reported with its error on all views
            Sm = min(Sn, 8)
            if Sm > 1:
                gS = [torch.stack([g] + [S.make_images(4, h, 512, seed=300 + 10 * b + gi)[0].to(device) for b in range(1, Sm)])
                      for gi, (g, h) in enumerate(zip(groups, MIXED_H))]
                inflight = {"scenes": Sm, "views_per_step": 20 * Sm, "modes": []}
                ref_r = None
                for fp8 in ((False, True) if fp8_built else (False,)):
                    enc.attention_fp8 = dec.attention_fp8 = fp8
                    fnm = lambda: run_scenes_mixed(enc, dec, gS)  # noqa: E731
                    om = fnm()
                    d = timed(fnm, max(2, ksteps // 2))
                    mode = {"dtype": args.precision + (" + fp8 attention" if fp8 else ""), "value": round(20 * Sm * max(2, ksteps // 2) / d, 2),
                            "ms_per_step": round(d / max(2, ksteps // 2) * 1e3, 3), "mfma_frac": round(Sm * fl * max(2, ksteps // 2) / d / 1e12 / 2500.0, 4)}
                    if not fp8:
                        ref_r = [r.clone() for r in om["render"]]
                        ref_u = [u.clone() for u in om["update"]]
                    else:
                        mode["render_rel_inf_vs_16bit_path_worst_view"] = max(float((a[b, v] - r[b, v]).abs().max() / r[b, v].abs().max())
                                                                             for a, r in zip(om["render"], ref_r) for b in range(Sm) for v in range(a.shape[1]))
                        mode["update_rel_inf_vs_16bit_path_worst_view"] = max(float((a[b] - r[b]).abs().max() / r[b].abs().max())
                                                                             for a, r in zip(om["update"], ref_u) for b in range(Sm))
                    inflight["modes"].append(mode)
                enc.attention_fp8 = dec.attention_fp8 = False
                if fp8_built:
                  v16, v8 = inflight["modes"][0]["value"], inflight["modes"][1]["value"]
                  inflight["verdict"] = (f"fp8 attention {'WINS' if v8 > v16 else 'LOSES'} with {Sm} mixed scenes in flight: {v8} vs {v16} views/s; its error vs the 16-bit "
                                       f"path {inflight['modes'][1]['render_rel_inf_vs_16bit_path_worst_view']:.2e} (render) is outside the 1e-3 target -> off by default")
                mixed["scenes_in_flight"] = inflight
                del gS, ref_r, ref_u, om
            configs.append(mixed)
            del groups
            # configs[3] on one GPU: online streaming memory, every frame updates the memory (engine.run_video)
            F = args.stream_frames
            vimgs, vts = S.make_images(F, H, W, seed=7)
            vimgs, vts = vimgs.to(device), vts.to(device)
            fnv = lambda: run_video(enc, dec, vimgs, vts)  # noqa: E731
            memv, _, kfs = fnv()
            d = timed(fnv, 1)
            stream_cfg = {"config": f"configs[3] MUSt3R_512 {F}-frame online streaming memory 384x512 on ONE GPU (window 25, keyframe every 3rd, "
                                    "in-place eviction): encode + per-frame memory update",
                          "value": round(F / d, 2), "unit": "frames/s", "ms_per_step": round(d * 1e3, 2), "frames": F,
                          "keyframes": len(kfs), "final_memory_tokens": int(memv[0][0].shape[1]), "dtype": dtype_label}
            # r06 (SURVEY.md section 8f "later"): the same stream through the context-parallel decoder call with a world of ONE rank -- the memory "sharded" over one
            # rank, every layer's cross attention leaving its fp32 partial and merging the one slot (include/must3r_hip.h must3r_hip_cp).  What the machinery costs
            # before any link is crossed; the N > 1 line times the real exchange (view_sharded.stream_context_parallel).
            try:
                from must3r_amd.parallel import run_video_sharded as _rvs
                fncp = lambda: _rvs(enc, dec, vimgs, vts, render=False, frame_counts=[F], context_parallel=True)  # noqa: E731
                ocp = fncp()
                dcp = timed(fncp, 1)
                stream_cfg["context_parallel_world1"] = {"value": round(F / dcp, 2), "unit": "frames/s", "ms_per_step": round(dcp * 1e3, 2),
                                                         "exchanges": int(ocp["cp_exchanges"]), "partial_bytes_per_exchange": int(ocp["cp_bytes_gathered"] // max(1, ocp["cp_exchanges"])),
                                                         "what": "run_video_sharded(context_parallel=True) without a process group: 12 partial merges + 12 slot merges per frame, no link"}
                del ocp
            except Exception as e:   # noqa: BLE001 -- a side figure must not take the line down
                stream_cfg["context_parallel_world1"] = {"error": repr(e)[:300]}
            configs.append(stream_cfg)
            del vimgs, memv
            # configs[1]: MUSt3R_224, 10 views of 224x224, one scene at a time and S scenes in flight
            e2, d2, _, _ = build_models(MUST3R_224, args.precision, device)
            i2 = torch.stack([S.make_images(10, 224, 224, seed=b)[0] for b in range(Sn)]).to(device)
            t2 = S.make_images(10, 224, 224, seed=0)[1]
            fl2 = scene_flops(196, 10, 10)
            fn2 = lambda: run_scene(e2, d2, i2[0], t2)  # noqa: E731
            fn2(); fn2()
            d = timed(fn2, 2 * ksteps)
            c1 = {"config": "configs[1] MUSt3R_224 10-view 224x224 scene (encode + update[2,1..] + render + activation)",
                  "value": round(10 * 2 * ksteps / d, 2), "unit": "views/s", "ms_per_step": round(d / (2 * ksteps) * 1e3, 3),
                  "scene_tflop": round(fl2 / 1e12, 3), "mfma_frac": round(fl2 * 2 * ksteps / d / 1e12 / 2500.0, 4),
                  "dtype": dtype_label, "note": "one scene at a time: launch-latency-bound at this size (0.94 ms at MFMA peak)"}
            if Sn > 1:
                fn2s = lambda: run_scenes(e2, d2, i2, t2)  # noqa: E731
                fn2s(); fn2s()
                d = timed(fn2s, ksteps)
                c1["scenes_in_flight"] = {"scenes": Sn, "value": round(Sn * 10 * ksteps / d, 2), "ms_per_step": round(d / ksteps * 1e3, 3),
                                          "mfma_frac": round(Sn * fl2 * ksteps / d / 1e12 / 2500.0, 4)}
            configs.append(c1)
            del e2, d2, i2
        else:
            # the view-SHARDED forms of ONE 20-view scene (must3r_amd/parallel.py): RCCL all-gather of the encoded keyframe tokens over xGMI,
            # sequential update replicated on every rank, encode and render view-sharded
            gidx = torch.arange(rank * V, (rank + 1) * V)
            keyframes = (gidx % world == 0)                    # weak form: every rank owns 20 views, 20 keyframes spread over all ranks
            kf_counts = [int((torch.arange(r * V, (r + 1) * V) % world == 0).sum()) for r in range(world)]   # static: no count exchange
            wimgs = S.make_images(V, H, W, seed=rank)[0].to(device)
            fnw = lambda: run_scene_sharded(enc, dec, wimgs, ts, keyframes, comm_dtype=tdt, keyframe_counts=kf_counts)  # noqa: E731
            fnw()
            d = timed(fnw, ksteps)
            kmax = max(kf_counts)
            esz = 2 if tdt in (torch.float16, torch.bfloat16) else 4
            rccl = {"world_size": world, "backend": dist.get_backend(), "collective": "all_gather_into_tensor (ONE per scene: the encoded keyframe tokens + one row of true-shape digits)",
                    "all_gather_bytes_per_step": {"weak": world * kmax * (N + 1) * cfg.enc_dim * esz}}
            sharded = {"weak": {"value": round(V * world * ksteps / d, 2), "unit": "views/s", "ms_per_step": round(d / ksteps * 1e3, 3), "views_per_step": V * world,
                                "keyframes": sum(kf_counts)}}
            configs.append({"config": f"configs[2] view-sharded, WEAK: every rank owns {V} views ({V * world} per scene), memory from {V} keyframes (every "
                                      f"{world}-th view): all-gather of the keyframe tokens [{backend}], replicated update, sharded encode / render",
                            "value": round(V * world * ksteps / d, 2), "unit": "views/s", "ms_per_step": round(d / ksteps * 1e3, 3),
                            "scaling": "weak", "dtype": dtype_label})
            del wimgs
            simgs, sts = S.make_images(V, H, W, seed=0)
            lo, hi = shard_range(V, rank, world)
            simgs, sts = simgs[lo:hi].to(device), sts[lo:hi]
            kf_all = torch.ones(hi - lo, dtype=torch.bool)
            kc_all = [shard_range(V, r, world)[1] - shard_range(V, r, world)[0] for r in range(world)]
            fns = lambda: run_scene_sharded(enc, dec, simgs, sts, kf_all, comm_dtype=tdt, keyframe_counts=kc_all)  # noqa: E731
            fns()
            d = timed(fns, ksteps)
            rccl["all_gather_bytes_per_step"]["strong"] = world * max(kc_all) * (N + 1) * cfg.enc_dim * esz
            sharded["strong"] = {"value": round(V * ksteps / d, 2), "unit": "views/s", "ms_per_step": round(d / ksteps * 1e3, 3), "views_per_step": V}
            configs.append({"config": f"configs[2] view-sharded, STRONG: the same {V}-view 384x512 scene sharded over {world} ranks "
                                      "(encode + render view-sharded, all-gather of the encoded tokens, sequential update replicated)",
                            "value": round(V * ksteps / d, 2), "unit": "views/s", "ms_per_step": round(d / ksteps * 1e3, 3),
                            "scaling": "strong", "dtype": dtype_label})
            # configs[3]: 200-frame stream sharded over the ranks
            F = args.stream_frames
            vimgs, vts = S.make_images(F, H, W, seed=7)
            lo, hi = shard_range(F, rank, world)
            vimgs, vts = vimgs[lo:hi].to(device), vts[lo:hi]
            fc_all = [shard_range(F, r, world)[1] - shard_range(F, r, world)[0] for r in range(world)]
            fnv = lambda: run_video_sharded(enc, dec, vimgs, vts, comm_dtype=tdt, render=False, frame_counts=fc_all)  # noqa: E731
            fnv()
            d = timed(fnv, 1)
            rccl["all_gather_bytes_per_step"]["stream"] = world * max(fc_all) * (N + 1) * cfg.enc_dim * esz
            sharded["stream"] = {"value": round(F / d, 2), "unit": "frames/s", "ms_per_step": round(d * 1e3, 2), "frames": F}
            configs.append({"config": f"configs[3] MUSt3R_512 {F}-frame online streaming memory, frames sharded over {world} ranks "
                                      "(encode sharded, all-gather of all frame tokens, per-frame memory update replicated)",
                            "value": round(F / d, 2), "unit": "frames/s", "ms_per_step": round(d * 1e3, 2), "frames": F,
                            "scaling": "strong", "dtype": dtype_label})
            # r06: ... and with the memory itself sharded over the ranks, the per-frame cross attention context-parallel (12 all-gathers of fp32 partials per frame)
            try:
                fncp = lambda: run_video_sharded(enc, dec, vimgs, vts, comm_dtype=tdt, render=False, frame_counts=fc_all, context_parallel=True)  # noqa: E731
                ocp = fncp()
                dcp = timed(fncp, 1)
                sharded["stream_context_parallel"] = {"value": round(F / dcp, 2), "unit": "frames/s", "ms_per_step": round(dcp * 1e3, 2), "frames": F,
                                                      "exchanges": int(ocp["cp_exchanges"]), "bytes_gathered": int(ocp["cp_bytes_gathered"]),
                                                      "memory_rows_per_rank": [int(r) for r in ocp["rows_per_rank"]]}
                rccl["all_gather_bytes_per_step"]["stream_context_parallel"] = int(ocp["cp_bytes_gathered"])
                del ocp
            except Exception as e:   # noqa: BLE001
                sharded["stream_context_parallel"] = {"error": repr(e)[:300]}
            del vimgs

    cpu_baseline, parity, torch_rocm_baseline = None, None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the oracle (a port of the reference's CPU path) on the metric's OWN unit of work -- one full 20-view scene: encode 20,
        # memory update [2,1,...,1], render 20 -- in its own process, 16 threads (the fastest count measured on the box), hard time limit.  It is pinned on the
        # committed real-reference fixture in the same pass (oracle_vs_reference_fixture), and the HIP pointmaps of ALL
        # views (update and render) are compared with it.
        import subprocess
        import tempfile
        import numpy as np
        ncores = os.cpu_count() or 1
        # 16 threads: the best count on the GPU box's host (profiles/r04_cpu_threads.txt: the 2-view scene takes 2.1 / 3.9 / 8.3 / 20.1 s at
        # 16 / 32 / 64 / 128 threads -- torch's CPU GEMMs lose to NUMA traffic and oversubscription beyond one socket's slice)
        threads = min(16, ncores)
        nv = min(args.cpu_views, V)
        with tempfile.TemporaryDirectory() as td:
            outp = os.path.join(td, "cpu.npz")
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--views", str(nv), "--H", str(H),
                                    "--W", str(W), "--threads", str(threads), "--check-fixture", "--out", outp],
                                   capture_output=True, text=True, timeout=args.cpu_timeout)
                info = json.loads(r.stdout.strip().splitlines()[-1])
                z = np.load(outp)
                ren_o, upd_o = torch.from_numpy(z["render"]), torch.from_numpy(z["update"])
                cpu_baseline = {"value": round(nv / info["seconds"], 4), "unit": "views/s", "cores": info["threads"],
                                "host_cores": ncores, "kind": "port",
                                "sample": f"one {nv}-view 384x512 scene (the unit the step runs S of): encode {nv} + memory update [2,1,...,1] + "
                                          f"render {nv} (fp32, torch CPU, SDPA attention)",
                                "seconds": round(info["seconds"], 2), "stages_s": {k: round(v, 2) for k, v in info["stages_s"].items()},
                                "oracle_vs_reference_fixture": info.get("oracle_vs_reference_fixture")}
                parity = {"views": nv, "reference": "CPU oracle (fp32) on the same seeded inputs; per-view = max over views of "
                                                     "||d_v||inf / ||ref_v||inf"}
                # what a must3r user gets on this MI355X today: the reference's path as eager PyTorch-ROCm (vendor GEMM / SDPA kernels) under the reference's own
                # precision policy -- the oracle port on cuda:0 (oracle/gpu_baseline.py), own process, ONE scene, never part of `value`
                try:
                    rg = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gpu_baseline.py"), "--views", str(nv), "--H", str(H), "--W", str(W),
                                         "--ref", outp], capture_output=True, text=True, timeout=300)
                    gi = json.loads(rg.stdout.strip().splitlines()[-1])
                    torch_rocm_baseline = {"value": round(gi["views_per_s"], 2), "unit": "views/s", "kind": "port on GPU",
                                           "what": "the oracle port (oracle/must3r_ref.py with torch leaves) as eager PyTorch-ROCm on the same GPU, one scene at a time, in the "
                                                   "reference's call pattern (one encoder call per view, one decoder call per schedule step, one render call per view)",
                                           "precision_policy": gi["precision_policy"], "attention": gi["attention"], "torch": gi["torch"],
                                           "seconds": round(gi["seconds"], 3), "stages_s": {k: round(v, 3) for k, v in gi["stages_s"].items()},
                                           "error_vs_fp32_cpu_oracle": gi.get("vs_fp32_cpu_oracle"),
                                           "compare_with": "value_single_scene (the HIP path, one scene at a time); never the target"}
                except Exception as e:
                    torch_rocm_baseline = {"value": None, "kind": "port on GPU", "error": repr(e)[:300]}

                def rel(a, b):
                    return float((a - b).abs().max() / b.abs().max())

                def cmp(ren, upd):
                    d = ren - ren_o
                    rv = [rel(ren[v], ren_o[v]) for v in range(nv)]
                    uv = [rel(upd[v], upd_o[v]) for v in range(nv)]
                    return {"pointmap_max_abs_err": float(max(d.abs().max(), (upd - upd_o).abs().max())), "rel_inf": rel(ren, ren_o),
                            "rel_l2": float(d.norm() / ren_o.norm()), "update_rel_inf": rel(upd, upd_o),
                            "render_per_view_max": max(rv), "update_per_view_max": max(uv),
                            "render_worst_view": int(max(range(nv), key=lambda v: rv[v])), "update_worst_view": int(max(range(nv), key=lambda v: uv[v])),
                            "update_last_view": rel(upd[nv - 1], upd_o[nv - 1])}
                from must3r_amd import _lib as _m3r_lib2
                for prec in ("fp16wa", "fp16w2", "fp16", "bf16") + (("fp16wa+fp8attn",) if _m3r_lib2.has_fp8_attention() else ()):
                    enc.precision = dec.precision = prec.split("+")[0]
                    enc.attention_fp8 = dec.attention_fp8 = prec.endswith("fp8attn")
                    out = run_scene(enc, dec, imgs[:nv], ts[:nv])
                    parity[prec] = cmp(out["render"].cpu(), out["update"].cpu())
                enc.precision = dec.precision = args.precision
                enc.attention_fp8 = dec.attention_fp8 = False
                if Sn > 1 and nv == V:
                    # the step itself: scene 0 of the batch IS the oracle's scene -> all its views against the oracle; every other scene
                    # against its own single-scene HIP run (tests/test_zz_batch_gpu.py checks every scene of a batch against the oracle
                    # at sizes the oracle finishes in seconds)
                    outS = step()
                    pf = cmp(outS["render"][0].cpu(), outS["update"][0].cpu())
                    others = []
                    for b in range(1, Sn):
                        one = run_scene(enc, dec, scenes[b], ts)
                        others.append(max(rel(outS["render"][b].cpu(), one["render"].cpu()), rel(outS["update"][b].cpu(), one["update"].cpu())))
                    pf["other_scenes_vs_their_single_scene_run_rel_inf"] = [round(o, 7) for o in others]
                    parity["step_scene0_of_" + str(Sn)] = pf
                    del outS
            except Exception as e:  # timeout or failure: report, never hang the bench
                cpu_baseline = {"value": None, "error": repr(e)[:300], "kind": "port"}

    if rank == 0:
        flops = Sn * world * scene_flops(N, V, V)
        # second half of BASELINE.json's metric ("...; pointmap max-abs-err vs ref"), inside `config` so that the driver's record keeps it: the BENCHED step's own
        # scene 0 (all views, all pixels) against the fp32 CPU oracle of the same scene, which this run pinned on the real-reference fixture (cpu_baseline)
        pkey = "step_scene0_of_" + str(Sn)
        pstep = (parity or {}).get(pkey) or (parity or {}).get(args.precision)
        parity_cfg = None
        if pstep:
            parity_cfg = {"rel_inf_worst_view": round(max(pstep["render_per_view_max"], pstep["update_per_view_max"]), 7),
                          "render_rel_inf_worst_view": round(pstep["render_per_view_max"], 7), "update_rel_inf_worst_view": round(pstep["update_per_view_max"], 7),
                          "max_abs_err": round(pstep["pointmap_max_abs_err"], 6), "tolerance": 1.0e-3, "asserted_in_tests": 8.0e-4,
                          "of": (f"scene 0 of the {Sn}-scene step this line times" if pkey in (parity or {}) else "one scene at a time") + ", 20 update + 20 render views, every pixel",
                          "vs": "fp32 CPU oracle (oracle/must3r_ref.py) on the same seeded inputs, itself " +
                                (f"{max((cpu_baseline or {}).get('oracle_vs_reference_fixture', {}).get('update_rel_inf', 0), (cpu_baseline or {}).get('oracle_vs_reference_fixture', {}).get('render_rel_inf', 0)):.1e}"
                                 if (cpu_baseline or {}).get("oracle_vs_reference_fixture") else "1-3e-6") +
                                " from the real reference's outputs (tests/golden/must3r512_v20.npz); the same scene against the real-reference fixtures, all pixels of "
                                "the worst views included, is asserted in tests/test_zz_r04_gpu.py",
                          "normalisation": "per view: ||d_v||inf / ||ref_v||inf over all pixels and channels of the raw [H, W, 7] head output"}
        s20 = next((r["value"] for r in (sweep or []) if r.get("scenes_in_flight") == 20), round(value, 2) if Sn == 20 else None)
        line = {
            "metric": "views/sec (whole node) MUSt3R_512 20-view 512x384; pointmap max-abs-err vs ref",
            "value": round(value, 2), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # BASELINE.md section 2 (r04 wording): `value` = whole-GPU throughput with S independent scenes in flight; `value_single_scene` = the same
            # scene ONE AT A TIME (V / wall-time(scene): the definition rounds 1-2 reported as `value` -- r01 243, r02 285, r03 322)
            "value_single_scene": (single or {}).get("value") if Sn > 1 else round(value, 2),
            "value_definition": f"{Sn} independent 20-view scenes in flight per GPU ({views_per_step} views per step) / step time; value_single_scene: one scene at a time",
            # like for like with BENCH_r03 / r04: the same step with 20 scenes in flight, measured in this run (null when --no-alt / N > 1 skipped it)
            "value_r04_definition": s20,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_label, "data": "synthetic",
            "config": {"workload": f"MUSt3R_512 {V}-view 384x512 scenes, S={Sn} independent scenes IN FLIGHT per GPU ({views_per_step} views/step); one at a time: value_single_scene",
                       # (the driver's record keeps `config`: the definition of `value` and the like-for-like batch-1 figure travel here as well as at the top level)
                       "value_definition": f"{Sn} independent {V}-view scenes in flight per GPU / step time (every scene: encode {V} + update [2,1,..,1] + render {V} + activation)",
                       "value_single_scene": (single or {}).get("value") if Sn > 1 else round(value, 2),
                       "model": "MUSt3R_512 ViT-L encoder / ViT-B memory decoder, random init, feedback single_mlp, memory_mode kv",
                       "views_per_step": views_per_step, "scenes_in_flight_per_gpu": Sn, "views_per_scene": V, "keyframes_per_scene": V,
                       "H": H, "W": W, "ms_per_scene": round(dt / args.steps / Sn * 1e3, 3),
                       "parity": parity_cfg, "value_r04_definition": s20,
                       "parallelism": "single GPU" if world == 1 else f"{world} replicas (independent scenes per rank, no data-path collective; "
                                                                       f"barrier + max over ranks) [{backend}]"},
            "roofline": roofline, "roofline_attention": roofline_attention, "roofline_gemm": roofline_gemm, "cpu_baseline": cpu_baseline, "torch_rocm_baseline": torch_rocm_baseline,
            "parity_vs_cpu_oracle": parity,
            "kernel_classes": classes, "stages_ms": stages, "single_scene": single, "alt": alt, "scenes_in_flight_sweep": sweep, "configs": configs, "postprocess_cam": cam,
            "rccl": rccl, "view_sharded": sharded,
            "multi_gpu": ("this line is a 1-GPU run; no RCCL run of the N > 1 paths has happened in the build environment (one GPU per box): the "
                          "view-sharded path is covered by world-size-2 gloo tests and a 2-rank gloo dry run of this script" if world == 1 else
                          f"{world} ranks, backend {backend}"),
            "scene_tflop": round(scene_flops(N, V, V) / 1e12, 2),
            "end_to_end_mfma_frac": round(flops * args.steps / dt / 1e12 / PEAK_TFLOPS[args.precision], 4),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
