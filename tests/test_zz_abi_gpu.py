"""GPU (-m gpu): the C ABI driven from a plain C program (tests/c/drive_abi.c) -- create / load_weight / finalize / encode /
decode with n_scenes = 2 / bounds-check refusal / render -- against the Python-driven path on the same weights and images.
SURVEY.md section 8(b) row "C ABI"; VERDICT r02 item 7."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from must3r_amd import _lib
from must3r_amd import synthetic as S
from must3r_amd.config import SMALL
from test_model_gpu import build

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_drives_encode_and_batched_decode(tmp_path):
    cc = shutil.which("gcc")
    if cc is None:
        pytest.skip("no gcc on this box")
    cfg = SMALL
    B, V, H, W = 2, 3, 224, 224
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    with open(tmp_path / "w.bin", "wb") as f:
        for pfx, sd in (("encoder.", sde), ("decoder.", sdd)):
            for k, v in sd.items():
                name = (pfx + k).encode()
                f.write(struct.pack("<i", len(name)) + name + struct.pack("<i", v.dim()) + struct.pack(f"<{v.dim()}q", *v.shape))
                f.write(v.detach().float().contiguous().numpy().tobytes())
    imgs = torch.stack([S.make_images(V, H, W, 70 + b)[0] for b in range(B)])
    (tmp_path / "img.bin").write_bytes(imgs.numpy().tobytes())
    exe = tmp_path / "drive_abi"
    lib_dir = os.path.join(ROOT, "must3r_amd")
    subprocess.run([cc, "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "tests", "c", "drive_abi.c"), "-o", str(exe), "-L", lib_dir, "-lmust3r_hip", "-L", "/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    args = [str(exe), str(tmp_path / "w.bin"), str(tmp_path / "img.bin"), str(tmp_path / "out.bin"), B, V, H, W, _lib.F16_WA, cfg.img_size,
            cfg.enc_dim, cfg.enc_depth, cfg.enc_heads, cfg.dec_dim, cfg.dec_depth, cfg.dec_heads]
    r = subprocess.run([str(a) for a in args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr)
    out = torch.from_numpy(np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(2, B, V, H, W, 7))
    # the same calls through the nn.Module mirrors (ctypes): bit-identical -- one library, one launch sequence
    enc, dec = build(cfg, "fp16wa")
    ts = S.make_images(V, H, W, 0)[1]
    x, pos = enc(imgs.reshape(B * V, 3, H, W).cuda(), ts.repeat(B, 1))
    x, pos = x.view(B, V, *x.shape[1:]), pos.view(B, V, *pos.shape[1:])
    t = ts.unsqueeze(0).expand(B, -1, -1)
    mem, upd = dec(x, pos, t, None)
    _, ren = dec(x, pos, t, mem, render=True)
    assert torch.equal(out[0], upd.cpu()) and torch.equal(out[1], ren.cpu())
