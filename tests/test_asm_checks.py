"""Static check of the kernels that count `lgkmcnt` by hand (gemm256k_kernel, attn4_kernel): scripts/checks/asm_inflight_regs.py walks the
generated gfx950 assembly and fails if any register of an in-flight LDS read is touched before the wait that covers it (the register
allocator is free to copy a fragment at a join or reuse its register -- it did, twice, while these kernels were written).  hipcc
cross-compiles without a GPU; the flags are the Makefile's."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# (attn4_kernel is the parked e4m3 attention kernel: compiled -- and checked -- only with -DM3R_ATTN_FP8, include/must3r_hip.h)
CASES = [("gemm.hip", ["-ffp-contract=off"], "gemm256k_kernel"), ("attention.hip", ["-fno-slp-vectorize", "-DM3R_ATTN_FP8"], "attn4_kernel")]


@pytest.mark.parametrize("src,extra,pattern", CASES)
def test_no_in_flight_register_is_touched(tmp_path, src, extra, pattern):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, "-S", "--cuda-device-only",
           os.path.join(ROOT, "must3r_amd", "csrc", src), "-o", str(out)]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "checks", "asm_inflight_regs.py"), str(out), pattern],
                       capture_output=True, text=True, timeout=600)
    kernels = [l for l in r.stdout.splitlines() if pattern in l]
    assert kernels, "no kernel matched " + pattern
    assert r.returncode == 0 and all(": OK" in l for l in kernels), r.stdout[-2000:]
    if src == "gemm.hip":
        # r06: the fold256 consumer prologues request their statistics by inline-asm loads (gemm.hip fold256_issue): the destination registers must not be touched
        # before the prologue's counted wait has retired them (scripts/checks/fold256_regs.py)
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "checks", "fold256_regs.py"), str(out)], capture_output=True, text=True, timeout=600)
        assert r2.returncode == 0 and "0 bad" in r2.stdout and " 4 fold256" in ("\n " + r2.stdout.splitlines()[-1]), r2.stdout[-2000:]
