"""CPU emulation of fp8 attention operand formats on the oracle (test/experiment infrastructure, never imported by the product).

Question: which operand quantisation lets the fp8 (MX-scaled e4m3) attention path meet a stated tolerance?  GEMMs stay fp32 here
(their fp16w2 error, ~4e-4, is known); only the four attention operands Q, K, V, P are quantised.

  python scripts/emul/fp8_attention.py [small|m224]
"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import must3r_ref as R
from must3r_amd import synthetic as S
from must3r_amd.config import SMALL, MUST3R_224

E4 = torch.float8_e4m3fn


def q8(t):
    return t.clamp(-448, 448).to(E4).float()


def q8_block(t, dim, block=32, target=8):
    """MX: one power-of-two scale per `block` elements along `dim` (E8M0), elements e4m3; amax -> [2^target, 2^(target+1))."""
    t = t.transpose(dim, -1)
    sh = t.shape
    n = sh[-1]
    pad = (-n) % block
    if pad:
        t = torch.nn.functional.pad(t, (0, pad))
    tb = t.reshape(*t.shape[:-1], -1, block)
    amax = tb.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(amax)) - target
    sc = torch.exp2(e)
    out = (q8(tb / sc) * sc).reshape(*t.shape)[..., :n]
    return out.reshape(sh).transpose(dim, -1)


def make_attn(mode):
    def attn(q, k, v, opq=None, sdpa=False):
        scale = q.shape[-1] ** -0.5
        qs = q * (scale * math.log2(math.e))     # the HIP path folds scale*log2(e) into q before rounding
        if mode["qk"] == "raw":
            qq, kk = q8(qs), q8(k)
        elif mode["qk"] == "mx":     # per (row, 32 d) block scales
            qq, kk = q8_block(qs, -1), q8_block(k, -1)
        elif mode["qk"] == "row":    # per row scale
            qq, kk = q8_block(qs, -1, 64), q8_block(k, -1, 64)
        elif mode["qk"] == "f16":
            qq, kk = qs.half().float(), k.half().float()
        else:
            qq, kk = qs, k
        s = qq @ kk.transpose(-2, -1)            # log2 domain
        m = s.amax(dim=-1, keepdim=True)
        p = torch.exp2(s - m)
        l = p.sum(dim=-1, keepdim=True)
        if mode["p"] == "e4":
            pp = q8(p * mode.get("pscale", 1.0)) / mode.get("pscale", 1.0)
            if mode.get("lsum_rounded", False):
                l = pp.sum(dim=-1, keepdim=True)
        elif mode["p"] == "f16":
            pp = p.half().float()
        else:
            pp = p
        if mode["v"] == "raw":
            vv = q8(v)
        elif mode["v"] == "mx":      # per (32 keys, d) block scales: blocks along the key dim
            vv = q8_block(v, -2)
        elif mode["v"] == "split":   # V = V_hi + V_lo, both e4m3 (MX block scales along the keys): two PV products
            vh = q8_block(v, -2)
            vv = vh + q8_block(v - vh, -2)
        elif mode["v"] == "split_raw":
            vh = q8(v)
            vv = vh + q8((v - vh) * 16.0) / 16.0
        elif mode["v"] == "f16":
            vv = v.half().float()
        else:
            vv = v
        return (pp @ vv) / l
    return attn


def run(cfgname, V=3, only=None):
    if cfgname == "small":
        cfg, H, W = SMALL, 224, 224
    else:
        cfg, H, W = MUST3R_224, 224, 224
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(V, H, W, 0)
    orig = R.softmax_attention
    with torch.no_grad():
        upd0, ren0, _ = R.run_scene(sde, sdd, cfg, imgs, ts, sdpa=False)
    modes = {
        "all raw e4m3 (r02 path)": dict(qk="raw", p="e4", v="raw"),
        "raw, P x32": dict(qk="raw", p="e4", v="raw", pscale=32.0),
        "QK only raw": dict(qk="raw", p="f32", v="f32"),
        "QK only mx32": dict(qk="mx", p="f32", v="f32"),
        "QK only row": dict(qk="row", p="f32", v="f32"),
        "P only e4": dict(qk="f32", p="e4", v="f32"),
        "P only e4 x32": dict(qk="f32", p="e4", v="f32", pscale=32.0),
        "P only e4 x32, l from rounded": dict(qk="f32", p="e4", v="f32", pscale=32.0, lsum_rounded=True),
        "V only raw": dict(qk="f32", p="f32", v="raw"),
        "V only mx32(keys)": dict(qk="f32", p="f32", v="mx"),
        "mx all, P x32": dict(qk="mx", p="e4", v="mx", pscale=32.0),
        "mx all, P x32, l rounded": dict(qk="mx", p="e4", v="mx", pscale=32.0, lsum_rounded=True),
        "QK f16, P e4 x32, V mx": dict(qk="f16", p="e4", v="mx", pscale=32.0),
        "QK mx, P f16, V f16": dict(qk="mx", p="f16", v="f16"),
        "QK raw, P f16, V f16": dict(qk="raw", p="f16", v="f16"),
        "QK raw, P e4 x32 l-rounded, V split": dict(qk="raw", p="e4", v="split", pscale=32.0, lsum_rounded=True),
        "QK raw, P e4 x32 l-rounded, V split_raw": dict(qk="raw", p="e4", v="split_raw", pscale=32.0, lsum_rounded=True),
        "QK raw, P f16, V split_raw": dict(qk="raw", p="f16", v="split_raw"),
    }
    if only:
        modes = {k: v for k, v in modes.items() if any(o in k for o in only)}
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    for name, mode in modes.items():
        R.softmax_attention = make_attn(mode)
        try:
            with torch.no_grad():
                upd, ren, _ = R.run_scene(sde, sdd, cfg, imgs, ts, sdpa=False)
        finally:
            R.softmax_attention = orig
        print(f"{cfgname:6s} V={V} {name:40s} update {rel(upd, upd0):.3e}  render {rel(ren, ren0):.3e}", flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    run(sys.argv[1] if len(sys.argv) > 1 else "small", int(sys.argv[2]) if len(sys.argv) > 2 else 3, sys.argv[3:] or None)
