#!/usr/bin/env python3
"""r06: where the time of the one-scene-at-a-time pass goes BETWEEN its kernels.  Input: the kernel trace (csv) of
    rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python bench.py --scenes 1 --step-only --steps 3 --warmup 1
Every dispatch has a start and an end timestamp on the device clock; the pass is ONE dependent chain on one stream, so the gap between the end of a kernel and the start of
the next one is the seam the chain pays at that boundary (dispatch of the next grid + its kernel-argument loads; the host is ahead of the device throughout).  Prints, for
the one-view update calls (the launches between two consecutive `attn_combine` ... markers are not needed: the whole pass is summarised per kernel family), the share of
wall time spent inside kernels and in seams, the seam histogram, and the per-family kernel time."""
import collections
import csv
import glob
import re
import sys


def main(d):
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not files:
        print("no kernel_trace.csv under", d)
        return 1
    rows = []
    with open(files[0]) as fh:
        for r in csv.DictReader(fh):
            name = re.sub(r"\(.*", "", r["Kernel_Name"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    m3r = [r for r in rows if "m3r" in r[2]]
    if not m3r:
        print("no library kernels in the trace")
        return 1
    # the timed passes: drop everything before the LAST third of the library's launches' first im2col (one pass = one im2col launch per encoder chunk)
    starts = [i for i, r in enumerate(m3r) if "im2col" in r[2]]
    passes = []
    for a, b in zip(starts, starts[1:] + [len(m3r)]):
        passes.append(m3r[a:b])
    passes = [p for p in passes if len(p) > 1000]
    print(f"{len(m3r)} library launches in the trace; passes (launches, wall ms, kernel ms): "
          + ", ".join(f"({len(p)}, {(p[-1][1] - p[0][0]) / 1e6:.1f}, {sum(e - s for s, e, _ in p) / 1e6:.1f})" for p in passes))
    # bench.py --step-only runs: warm-up pass(es) (first-use weight packing), the timed passes, ONE pass under the library's HIP-event profiler (an event pair around
    # every launch: ~10 us per boundary), the parity pass.  The timed passes are the ones with the least wall time: take the three shortest.
    passes = sorted(passes, key=lambda p: p[-1][1] - p[0][0])[:3]
    print(f"the three shortest = the timed passes: {[len(p) for p in passes]} launches")
    tot_k = tot_gap = tot_wall = 0
    gaps = []
    fam = collections.defaultdict(lambda: [0, 0])
    for p in passes:
        tot_wall += p[-1][1] - p[0][0]
        for i, (s, e, n) in enumerate(p):
            tot_k += e - s
            f = re.sub(r"^_ZN3m3r\d+", "", n)[:60]
            fam[f][0] += e - s
            fam[f][1] += 1
            if i:
                g = s - p[i - 1][1]
                gaps.append(g)
                tot_gap += max(g, 0)
    n = len(passes)
    print(f"per pass: wall {tot_wall / n / 1e6:.2f} ms = kernels {tot_k / n / 1e6:.2f} ms + seams {tot_gap / n / 1e6:.2f} ms ({100.0 * tot_gap / tot_wall:.1f} % of the pass); "
          f"{len(gaps) // n} boundaries, mean seam {tot_gap / max(1, len(gaps)) / 1e3:.2f} us, mean kernel {tot_k / max(1, len(gaps) + n) / 1e3:.2f} us")
    gs = sorted(gaps)
    q = lambda f: gs[min(len(gs) - 1, int(f * len(gs)))] / 1e3  # noqa: E731
    print(f"seam quantiles (us): 10 % {q(0.1):.2f}  50 % {q(0.5):.2f}  90 % {q(0.9):.2f}  99 % {q(0.99):.2f}  max {gs[-1] / 1e3:.1f};  "
          f"seams above 5 us: {sum(1 for g in gs if g > 5000) // n} per pass worth {sum(g for g in gs if g > 5000) / n / 1e6:.2f} ms (the per-call view-table uploads: one per decoder call)")
    print("kernel family: launches per pass, ms per pass, mean us")
    for f, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"  {f:34s} {c // n:6d} {t / n / 1e6:8.2f} {t / c / 1e3:8.2f}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
