"""CPU: host-side logic of the product package and the C-ABI surface (no compute calls -- no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from must3r_amd import _lib, synthetic as S
from must3r_amd.config import TINY, MUST3R_512
from must3r_amd.engine import demo_mem_batches
import must3r_amd.model as M


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "must3r_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(must3r_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert set(declared) == set(_lib.EXPORTS), set(declared) ^ set(_lib.EXPORTS)
    assert os.path.exists(_lib.LIB_PATH), "build the extension first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().must3r_hip_abi_version() == _lib.ABI_VERSION


def test_rope_table_host_entry_point():
    lib = _lib.load()
    buf = (ctypes.c_float * (8 * 32))()
    assert lib.must3r_hip_rope_table(100.0, 1.0, 8, buf) == 0
    t = torch.tensor(list(buf)).view(8, 16, 2)
    from oracle import must3r_ref as R
    cos, sin = R.rope_tables(8, 100.0, 1.0, 32)
    assert torch.allclose(t[..., 0], cos, atol=1e-6) and torch.allclose(t[..., 1], sin, atol=1e-6)


def test_create_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.HipError):
        _lib.Context(TINY, 0)


def test_modules_have_reference_state_dict_keys_and_refuse_cpu():
    cfg = TINY
    enc = M.Dust3rEncoder(img_size=(cfg.img_size,) * 2, embed_dim=cfg.enc_dim, depth=cfg.enc_depth, num_heads=cfg.enc_heads)
    dec = M.MUSt3R(img_size=(cfg.img_size,) * 2, enc_embed_dim=cfg.enc_dim, embed_dim=cfg.dec_dim, depth=cfg.dec_depth,
                   num_heads=cfg.dec_heads, feedback_type="single_mlp", memory_mode="kv", mem_dropout=0.1,
                   use_xformers_mask=True)  # training kwargs are swallowed like decoder.py:37
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    assert set(enc.state_dict()) == set(sde) and set(dec.state_dict()) == set(sdd)
    for k, v in enc.state_dict().items():
        assert tuple(v.shape) == tuple(sde[k].shape), k
    enc.load_state_dict(sde, strict=True)
    dec.load_state_dict(sdd, strict=True)
    assert float(dec.feedback_layer.fc2.weight.abs().sum()) > 0
    assert enc.patch_size == 16 and enc.embed_dim == cfg.enc_dim and dec.memory_mode == "kv"
    assert M.get_pointmaps_activation(dec, verbose=False) == M.ActivationType.NORM_EXP
    imgs, ts = S.make_images(1, 32, 32, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(imgs, ts)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dec(torch.zeros(1, 1, 4, cfg.enc_dim), torch.zeros(1, 1, 4, 2, dtype=torch.int64), torch.tensor([[[32, 32]]]))
    dec.change_memory_mode("norm_y")
    assert all(b.memory_mode == "norm_y" for b in dec.blocks_dec)


def test_full_size_key_contract():
    # key names / shapes of the released geometry without allocating the model
    sde = S.make_encoder_state_dict(TINY, 0)
    assert "patch_embed.proj.weight" in sde and "blocks_enc.1.mlp.fc2.bias" in sde and "norm_enc.weight" in sde
    assert MUST3R_512.output_dim == 1792 and MUST3R_512.enc_dim // MUST3R_512.enc_heads == 64


def test_arg_rewriting():
    s = M.convert_decoder_args("CausalMUSt3R(img_size=(512, 512), feedback_type='single_mlp', memory_mode=\"kv\")")
    assert s == "MUSt3R(img_size=(512,512),feedback_type='single_mlp',memory_mode=\"kv\",landscape_only=False)"
    s2 = M.set_image_size_in_args("MUSt3R(img_size=(224, 224))", 512, verbose=False)
    assert s2 == "MUSt3R(img_size=(512,512),pos_embed='RoPE100_224:512')"
    assert M.set_image_size_in_args("MUSt3R(img_size=(512,512),pos_embed='RoPE100')", 512, verbose=False) == \
        "MUSt3R(img_size=(512,512),pos_embed='RoPE100')"
    from must3r_amd.model.blocks import parse_pos_embed
    assert parse_pos_embed("RoPE100") == (100.0, 1.0) and parse_pos_embed("RoPE100_224:512") == (100.0, 224 / 512)


def test_load_model_roundtrip_on_cpu(tmp_path):
    cfg = TINY
    class A:  # noqa: E301
        encoder = f"Dust3rEncoder(img_size=({cfg.img_size},{cfg.img_size}),embed_dim={cfg.enc_dim},depth={cfg.enc_depth},num_heads={cfg.enc_heads})"
        decoder = (f"CausalMUSt3R(img_size=({cfg.img_size}, {cfg.img_size}), enc_embed_dim={cfg.enc_dim}, embed_dim={cfg.dec_dim}, "
                   f"depth={cfg.dec_depth}, num_heads={cfg.dec_heads}, feedback_type='single_mlp', memory_mode='kv', mem_dropout=0.1)")
    import argparse
    ns = argparse.Namespace(encoder=A.encoder, decoder=A.decoder)
    p = tmp_path / "ckpt.pth"
    torch.save({"args": ns, "encoder": S.make_encoder_state_dict(cfg, 1), "decoder": S.make_decoder_state_dict(cfg, 1)}, p)
    enc, dec = M.load_model(str(p), device="cpu", verbose=False)
    assert isinstance(enc, M.Dust3rEncoder) and isinstance(dec, M.MUSt3R) and not enc.training
    assert torch.equal(dec.image2_embed, S.make_decoder_state_dict(cfg, 1)["image2_embed"])
    enc2, dec2 = M.load_model(str(p), device="cpu", img_size=128, verbose=False)
    assert dec2.cfg.rope_f0 == 64 / 128 and dec2.cfg.img_size == 128


def test_demo_mem_batches():
    assert demo_mem_batches(20) == [2] + [1] * 18          # demo/inference.py:188-191 defaults
    assert demo_mem_batches(2) == [2] and demo_mem_batches(1) == [1]
    assert demo_mem_batches(7, 2, 2) == [2, 2, 2, 1]


def test_memory_surgery_helpers_match_reference_and_keep_buffers():
    """remove / restore-label / update-in-place helpers against the reference's own
    engine/inference.py:205-228 (when the tree is present), on CPU tensors; the in-place form must keep buffer ownership."""
    from must3r_amd.engine import remove_from_mem, restore_label_in_mem, update_in_mem
    from must3r_amd.model.decoder import _MemBuffers
    torch.manual_seed(0)
    depth, N, D = 3, 5, 8
    labels = torch.arange(4).repeat_interleave(N).view(1, -1)
    owner = _MemBuffers(depth, 64, D, torch.float32, torch.device("cpu"))
    for b in owner.bufs:
        b.normal_()
    owner.valid = 4 * N
    vals = owner.views(4 * N)
    ref_vals = [v.clone() for v in vals]

    def ref_remove(mem_values, mem_labels, idx):       # engine/inference.py:205-213
        keep = mem_labels != idx
        B, _, Dd = mem_values[0].shape
        return [m[keep].view(B, -1, Dd) for m in mem_values], mem_labels[keep].view(B, -1)

    got_v, got_l = remove_from_mem(vals, labels.clone(), 1)
    exp_v, exp_l = ref_remove(ref_vals, labels.clone(), 1)
    assert torch.equal(got_l, exp_l) and all(torch.equal(a, b) for a, b in zip(got_v, exp_v))
    assert owner.valid == 3 * N and all(getattr(v, "_m3r_owner", None) is owner for v in got_v)   # still appendable
    assert got_v[0].data_ptr() == owner.bufs[0].data_ptr()
    # foreign tensors (no owner): plain functional behaviour
    f_v, f_l = remove_from_mem([v.clone() for v in exp_v], exp_l.clone(), 3)
    e_v, e_l = ref_remove(exp_v, exp_l.clone(), 3)
    assert torch.equal(f_l, e_l) and all(torch.equal(a, b) for a, b in zip(f_v, e_v))
    assert torch.equal(restore_label_in_mem(torch.tensor([[0, 7, 7, 2]]), 1, 7), torch.tensor([[0, 1, 1, 2]]))
    new_vals = [torch.randn(1, 2 * N, D) for _ in range(depth)]
    new_labels = torch.tensor([[5] * N + [6] * N])
    upd = update_in_mem(got_v, new_vals, got_l, new_labels, 2, 6)
    assert torch.equal(upd[1][got_l == 2], new_vals[1][new_labels == 6]) and torch.equal(upd[1][got_l == 0], exp_v[1][exp_l == 0])
    from conftest import HAS_REFERENCE
    if HAS_REFERENCE:
        from oracle import ref_shims
        ref_shims.install()
        import must3r.engine.inference as RI
        r_v, r_l = RI._remove_from_mem([v.clone() for v in ref_vals], labels.clone(), 1)
        assert torch.equal(r_l, exp_l) and all(torch.equal(a, b) for a, b in zip(r_v, exp_v))
        assert torch.equal(RI._restore_label_in_mem(torch.tensor([[0, 7, 7, 2]]), 1, 7), torch.tensor([[0, 1, 1, 2]]))


def test_auxiliary_entry_points_refuse_cpu_tensors():
    """like the forward path, the keyframe test, the retrieval front-end and compute_cam have no CPU route."""
    import torch
    from must3r_amd import slam_nn, retrieval
    from must3r_amd.engine import postprocess
    with pytest.raises(RuntimeError):
        slam_nn.nn_distances(torch.zeros((4, 3)), torch.zeros((4, 3)))
    with pytest.raises(RuntimeError):
        slam_nn.get_searcher("kdtree-scipy-quadrant_x2").add_pts(torch.zeros((4, 3)), cam_center=torch.zeros(3))
    with pytest.raises(RuntimeError):
        retrieval.how_select_local(torch.zeros((1, 4, 8)), torch.zeros((1, 4)), 2)
    with pytest.raises(RuntimeError):
        retrieval.Whitener(8)(torch.zeros((1, 4, 8)))
    with pytest.raises(RuntimeError):
        postprocess(torch.zeros((1, 16, 16, 7)), compute_cam=True)
    assert slam_nn.get_searcher("none") is None
    with pytest.raises(ValueError):
        slam_nn.get_searcher("faiss")


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The drop-in boundary is a C ABI: include/must3r_hip.h must compile as C (no C++/torch types) and a C program must
    link against libmust3r_hip.so by the declared names (host-only entry points are called; no GPU)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "must3r_amd")
    if not os.path.exists(os.path.join(lib_dir, "libmust3r_hip.so")):
        pytest.skip("library not built")
    src = tmp_path / "abi.c"
    src.write_text('#include "must3r_hip.h"\n#include <stdio.h>\n'
                   'int main(void) {\n'
                   '  float t[2 * 16 * 2];   /* [npos][16][cos, sin] */\n'
                   '  if (must3r_hip_abi_version() != MUST3R_HIP_ABI_VERSION) return 2;\n'
                   '  if (must3r_hip_rope_table(100.0f, 1.0f, 2, t) != 0) return 3;\n'
                   '  must3r_hip_ctx* c = 0;\n'
                   '  if (must3r_hip_create(0, 0, &c) == 0) return 4;   /* null config must be refused */\n'
                   '  printf("%s|%.8f\\n", must3r_hip_last_error(), t[32]);\n'
                   '  return 0;\n}\n')
    exe = tmp_path / "abi"
    inc = os.path.join(root, "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", os.path.join(inc, "must3r_hip.h")], check=True)
    subprocess.run([gcc, "-std=c99", "-Wall", "-I", inc, str(src), "-o", str(exe), "-L", lib_dir, "-lmust3r_hip",
                    "-Wl,-rpath," + lib_dir], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    err, t16 = r.stdout.strip().split("|")
    import math
    assert "null" in err and abs(float(t16) - math.cos(1.0)) < 1e-6   # angle(p=1, i=0) = f0 = 1


def test_backend_switch_dispatch_without_the_reference(monkeypatch):
    """must3r_amd.backend on a stand-in `must3r.model` (no reference tree needed): install() wraps load_model and re-points early importers,
    the switch selects the loader, arguments travel unchanged, uninstall() restores everything."""
    import sys
    import types
    import must3r_amd.backend as hb
    import must3r_amd.model as HM
    calls = []

    def ref_load_model(chkpt_path, encoder=None, decoder=None, device="cuda", img_size=None, memory_mode=None, verbose=True):
        calls.append(("ref", chkpt_path, encoder, decoder, device, img_size, memory_mode, verbose))
        return "ref-enc", "ref-dec"

    pkg, mod, early = types.ModuleType("must3r"), types.ModuleType("must3r.model"), types.ModuleType("must3r.slam_like")
    mod.load_model = early.load_model = ref_load_model
    pkg.model = mod
    for name, m in (("must3r", pkg), ("must3r.model", mod), ("must3r.slam_like", early)):
        monkeypatch.setitem(sys.modules, name, m)
    monkeypatch.setattr(HM, "load_model", lambda *a: (calls.append(("hip",) + a), ("hip-enc", "hip-dec"))[1])
    try:
        assert hb.install() is mod and hb.install() is mod                    # idempotent
        assert mod.load_model is early.load_model and mod.load_model.__wrapped__ is ref_load_model
        assert mod.load_model("a.pth", device="cpu", img_size=224) == ("ref-enc", "ref-dec")
        mod.toggle_hip_backend(True)
        assert mod.is_hip_backend_enabled()
        assert early.load_model("b.pth", None, None, "cuda:1", 512, "kv", False) == ("hip-enc", "hip-dec")
        mod.toggle_hip_backend(False)
        assert mod.load_model("c.pth") == ("ref-enc", "ref-dec")
    finally:
        hb.uninstall()
    assert mod.load_model is ref_load_model and early.load_model is ref_load_model and not hasattr(mod, "toggle_hip_backend")
    assert calls == [("ref", "a.pth", None, None, "cpu", 224, None, True), ("hip", "b.pth", None, None, "cuda:1", 512, "kv", False),
                     ("ref", "c.pth", None, None, "cuda", None, None, True)]


def test_label_run_mirror_is_dropped_when_the_labels_are_edited_in_place():
    """engine._label_runs: the host mirror of the label layout is only trusted while the tensor's version counter is the one it
    was attached at -- an in-place edit outside the helpers (the reference's own _restore_label_in_mem, user code) makes it stale,
    and a stale mirror would evict / overwrite the wrong rows."""
    from must3r_amd import engine as E
    labels = torch.tensor([[0, 0, 1, 1, 2, 2]])
    E.attach_label_runs(labels, [(0, 2), (1, 2), (2, 2)])
    assert E._label_runs(labels) == [(0, 2), (1, 2), (2, 2)]
    vals = [torch.arange(6.0).view(1, 6, 1)]
    v2, l2 = E.remove_from_mem(vals, labels, 1)
    assert l2.tolist() == [[0, 0, 2, 2]] and E._label_runs(l2) == [(0, 2), (2, 2)] and v2[0].flatten().tolist() == [0, 1, 4, 5]
    labels[labels == 2] = 7                       # in place, behind the helpers' back
    assert E._label_runs(labels) is None          # stale mirror rejected: the helpers take the mask path
    v3, l3 = E.remove_from_mem(vals, labels, 7)
    assert l3.tolist() == [[0, 0, 1, 1]] and v3[0].flatten().tolist() == [0, 1, 2, 3]
    l4 = E.restore_label_in_mem(l2, 5, 2)         # the helper keeps its own mirror fresh
    assert l4.tolist() == [[0, 0, 5, 5]] and E._label_runs(l4) == [(0, 2), (5, 2)]


def test_epilogue_lane_exchange_gives_whole_line_stores():
    """Pure-numpy model of the 16-bit store path of epilogue_row (gemm.hip): after v_permlane32_swap, v_permlane16_swap and the DPP row rotation by 8
    under bank masks, lane l stores 16 bytes = 8 consecutive columns, eight lanes cover one 128-byte row segment, and the 64 lanes together
    cover every (row, column) of the 16 x 64 fragment block exactly once."""
    import numpy as np
    lanes = np.arange(64)
    fr, fg = lanes & 15, lanes >> 4
    # value held by lane l in fragment j, dword d (two 16-bit outputs): identify it by (row, col of its first output)
    def src(j, d):
        return np.stack((fr, j * 16 + fg * 4 + d * 2), -1)            # [64, 2]
    def permlane32_swap(a, b):   # a' = {a.lo32, b.lo32}, b' = {a.hi32, b.hi32}
        return np.concatenate((a[:32], b[:32])), np.concatenate((a[32:], b[32:]))
    def permlane16_swap(a, b):   # odd 16-lane rows of a <-> even rows of b
        a2, b2 = a.copy(), b.copy()
        for r in (1, 3):
            a2[r * 16:(r + 1) * 16] = b[(r - 1) * 16:r * 16]
            b2[(r - 1) * 16:r * 16] = a[r * 16:(r + 1) * 16]
        return a2, b2
    def dpp_ror8(old, srcv, bank_mask):   # row_ror:8 within each 16-lane row, written only to lanes whose 4-lane bank is enabled
        out = old.copy()
        for l in range(64):
            if bank_mask >> ((l & 15) >> 2) & 1:
                out[l] = srcv[(l & ~15) | ((l + 8) & 15)]
        return out
    o = []
    for q in range(2):
        regs = [None] * 4
        for d in range(2):
            s0, s1 = permlane32_swap(src(2 * q, d), src(2 * q + 1, d))
            t0, t1 = permlane16_swap(s0, s1)
            regs[d], regs[2 + d] = t0, t1
        o.append(regs)
    x = [dpp_ror8(o[0][d], o[1][d], 0xC) for d in range(4)]
    y = [dpp_ror8(o[1][d], o[0][d], 0x3) for d in range(4)]
    seen = set()
    for name, regs, row_add in (("x", x, 0), ("y", y, 8)):
        for l in range(64):
            row = (l & 7) + row_add
            col0 = ((l >> 5) & 1) * 16 + ((l >> 4) & 1) * 8 + ((l >> 3) & 1) * 32
            for d in range(4):
                r, c = regs[d][l]
                assert r == row and c == col0 + 2 * d, (name, l, d, (r, c), (row, col0 + 2 * d))
                seen.add((r, c)); seen.add((r, c + 1))
    assert len(seen) == 16 * 64


def test_copies_and_pickles_of_the_modules_are_independent(tmp_path):
    """nothing may be patched onto the (sub)modules -- `copy.deepcopy(model).half()` converts the COPY and only the copy,
    `torch.save(model)` works, and a conversion of a child moves the cheap eval-mode fingerprint of its owner."""
    import copy
    cfg = TINY
    dec = M.MUSt3R(img_size=(cfg.img_size, cfg.img_size), enc_embed_dim=cfg.enc_dim, embed_dim=cfg.dec_dim, depth=cfg.dec_depth,
                   num_heads=cfg.dec_heads).eval()
    fp0 = dec._fingerprint()
    assert dec._fingerprint() == fp0                                   # stable while nothing changes
    assert not any("_apply" in m.__dict__ or "_m3r_hooked" in m.__dict__ for m in dec.modules())
    cp = copy.deepcopy(dec).half()
    assert all(p.dtype == torch.float16 for p in cp.parameters())
    assert all(p.dtype == torch.float32 for p in dec.parameters())
    assert dec._fingerprint() == fp0                                   # the original did not move
    assert cp._ctx is None and cp._synced is None and "_param_cache" not in cp.__dict__
    # a child converted on its own: new storage -> the owner's fingerprint moves (and back to a different one after the round trip)
    dec.blocks_dec[1].attn.qkv.double()
    fp1 = dec._fingerprint()
    assert fp1 != fp0
    dec.blocks_dec[1].attn.qkv.float()
    # in-place edits: version counters
    fp2 = dec._fingerprint()
    # (the convert-and-back round trip of a CHILD is the documented blind spot when the allocator returns the same blocks -- _hip_module.py,
    # "NOT seen": refresh_weights() -- ; what IS guaranteed: the fingerprint is stable again, and an order-changing swap of two storages moves it)
    assert dec._fingerprint() == fp2
    a, b = dec.blocks_dec[0].norm1.weight, dec.blocks_dec[0].norm2.weight
    da, db = a.data, b.data
    a.data, b.data = db, da                                             # same multiset of pointers, versions and dtypes: only the ORDER moved
    assert dec._fingerprint() != fp2
    a.data, b.data = da, db
    assert dec._fingerprint() == fp2
    with torch.no_grad():
        dec.blocks_dec[0].mlp.fc1.bias.add_(1.0)
    assert dec._fingerprint() != fp2
    fp3 = dec._fingerprint()
    dec.blocks_dec[0].load_state_dict(dec.blocks_dec[0].state_dict())   # a child's load_state_dict = in-place copies
    assert dec._fingerprint() != fp3
    # pickling the whole module (the reference's checkpoints hold state dicts, but torch.save(model) is what ad-hoc scripts do)
    path = tmp_path / "dec.pt"
    torch.save(dec, path)
    back = torch.load(path, weights_only=False)
    assert back._ctx is None and back._synced is None
    assert all(torch.equal(a, b) for a, b in zip(back.state_dict().values(), dec.state_dict().values()))
    assert back._fingerprint() != dec._fingerprint()                    # its own storage


def test_memory_rows_of_the_other_attention_mode_are_refused_not_cast():
    """ADVICE r03 + r04: with attention_fp8 + 'kv' the memory rows are opaque uint8 [B, Nm, 3 D]; every other mode holds floating-point [B, Nm, mem_D]
    rows.  A memory of the OTHER format must be refused in BOTH directions (a numeric cast would be silently wrong), update and render alike: the
    check runs in front of both branches of `_forward_scene`."""
    cfg = TINY
    dec = M.MUSt3R(img_size=(cfg.img_size, cfg.img_size), enc_embed_dim=cfg.enc_dim, embed_dim=cfg.dec_dim, depth=cfg.dec_depth,
                   num_heads=cfg.dec_heads, memory_mode="kv").eval()
    D, Nm = cfg.dec_dim, 8
    mem16 = [torch.zeros((1, Nm, 2 * D), dtype=torch.float16) for _ in range(cfg.dec_depth)]
    mem8 = [torch.zeros((1, Nm, 3 * D), dtype=torch.uint8) for _ in range(cfg.dec_depth)]
    dec._check_memory_rows(mem16, 2 * D, False)                       # the right format passes, both ways
    dec._check_memory_rows(mem8, 3 * D, True)
    dec._check_memory_rows([m.float() for m in mem16], 2 * D, False)   # a wider floating type is a legitimate numeric cast
    with pytest.raises(ValueError, match="attention_fp8 on holds opaque"):
        dec._check_memory_rows(mem8, 2 * D, False)                    # fp8 rows read with attention_fp8 off (the direction r04 missed)
    with pytest.raises(ValueError, match="attention_fp8 off cannot"):
        dec._check_memory_rows(mem16, 3 * D, True)                    # 16-bit rows read with attention_fp8 on
    with pytest.raises(ValueError):
        dec._check_memory_rows([torch.zeros((1, Nm, 3 * D), dtype=torch.float16)] * cfg.dec_depth, 2 * D, False)   # right dtype, wrong row width
    # the check sits in front of BOTH the render and the update branch
    src = open(os.path.join(ROOT, "must3r_amd", "model", "decoder.py")).read()
    i, j, k = src.index("self._check_memory_rows(mem_vals, mem_D, fp8_rows)"), src.index("        if render:\n            # read-only"), src.index("self._writable_memory(mem_vals, Nm")
    assert i < j < k


def test_set_option_is_validated(monkeypatch):
    """ABI 8: the library's A/B switches live in ONE validated table (csrc/options.hpp): an unknown name or a value outside the switch's range is refused with an
    error string; a refused call leaves the switch as it was.  (No GPU needed: nothing is launched.)"""
    from must3r_amd import _lib
    L = _lib.load()
    assert L.must3r_hip_set_option(b"PERSIST", 1) == 0
    for name, bad in ((b"PERSIST", 2), (b"PERSIST", -1), (b"GEMM256", 3), (b"ENC_CHUNK_ROWS", 0), (b"ENC_CHUNK_ROWS", 1 << 40), (b"NOPE", 1), (b"", 0)):
        assert L.must3r_hip_set_option(name, bad) == 1, (name, bad)
        msg = L.must3r_hip_last_error().decode()
        assert "set_option" in msg and (name.decode() in msg or not name), msg
    with pytest.raises(_lib.HipError):
        _lib.set_option("SPARSE_LO", 7)
    for name, ok in (("PERSIST", 0), ("PERSIST", 1), ("GEMM256", 2), ("GEMM256", 1), ("ENC_CHUNK_ROWS", 32768), ("ATTN_LZ", 1), ("LNFOLD", 1), ("SPARSE_LO", 1)):
        _lib.set_option(name, ok)



def test_causal_module_constructor_and_memory_tail():
    """r06: CausalMUSt3R is a forward now (must3r/model/decoder.py:352-553).  Its constructor takes the reference's arguments, refuses the training-time memory dropout
    and unknown dropout modes like the reference (decoder.py:376), and its tuple tail follows decoder.py:461-464."""
    import must3r_amd.model as M
    from must3r_amd.config import TINY as cfg
    kw = dict(img_size=(cfg.img_size,) * 2, enc_embed_dim=cfg.enc_dim, embed_dim=cfg.dec_dim, depth=cfg.dec_depth, num_heads=cfg.dec_heads,
              feedback_type="single_mlp", memory_mode="kv", landscape_only=False)
    dec = M.CausalMUSt3R(protected_imgs=2, use_xformers_mask=True, use_mem_mask=True, **kw)
    assert isinstance(dec, M.MUSt3R) and dec._causal and dec.protected_imgs == 2
    assert set(dec.state_dict().keys()) == set(M.MUSt3R(**kw).state_dict().keys())          # the class adds no parameters
    labels = torch.zeros((1, 36), dtype=torch.int64)
    assert dec._memory_tail([], labels, 3, 0, 0, 3, 12)[2:] == (3, 2, 24)                   # init with 3 views: 2 protected images = 24 tokens
    assert dec._memory_tail([], labels, 5, 2, 24, 2, 12)[2:] == (5, 2, 24)                  # later calls add none
    assert M.MUSt3R(**kw)._memory_tail([], labels, 3, 0, 0, 3, 12)[2:] == (3, 3, 36)        # MUSt3R: every image protected (decoder.py:336)
    with pytest.raises(NotImplementedError, match="mem_dropout"):
        M.CausalMUSt3R(mem_dropout=0.1, **kw)
    with pytest.raises(ValueError, match="dropout mode"):
        M.CausalMUSt3R(dropout_mode="sometimes", **kw)
    with pytest.raises(TypeError, match="no list dispatch"):
        dec([torch.zeros(1)], [torch.zeros(1)], [torch.zeros(1)])
