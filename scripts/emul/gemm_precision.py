"""CPU emulation of the GEMM operand formats of the HIP path on the oracle (experiment infrastructure, never imported by the product).

  python scripts/emul/gemm_precision.py massive          # outlier-channel K-extension on the massive-activation scene of tests/test_edge_gpu.py
  python scripts/emul/gemm_precision.py split [views]    # which weights need the W_hi + W_lo split for <= 8e-4 (MUSt3R_224)

Model of the path: every Linear = fp32 accumulation of products of 16-bit operands; activations fp16; weights fp16 (`plain`) or
fp16 hi + fp16 lo (`split`, ~fp32); attention operands q, k, v, P fp16; residual stream, LayerNorm, softmax fp32; head fp32.
Outlier extension: for the Linears that read a LayerNorm output, the channels whose |value| exceeds `thr` x the median channel
absmax (at most `cap` channels) also carry their fp16 low part (one extra K-tile in the real kernel).
"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import must3r_ref as R
from must3r_amd import synthetic as S
from must3r_amd.config import SMALL, MUST3R_224

h = lambda t: t.half().float()
LN_FED = ("attn.qkv.weight", "mlp.fc1.weight", "cross_attn.projq.weight", "cross_attn.projk.weight", "cross_attn.projv.weight",
          "feedback_layer.fc1.weight", "feat_embed_enc_to_dec.weight")


class Emu:
    def __init__(self, sds, split=lambda name: True, ext=None, act_split=False):
        self.names = {}
        for pfx, sd in sds:
            for k, v in sd.items():
                self.names[id(v)] = pfx + k
        self.split, self.ext, self.act_split = split, ext, act_split
        self.nflag = {}

    def linear(self, x, w, b, opq=None):
        name = self.names.get(id(w), "?")
        if name.endswith("head_dec.proj.weight") or name == "?":
            y = x @ w.t()
            return y + b if b is not None else y
        wh = h(w)
        wq = wh + h(w - wh) if self.split(name) else wh
        xh = h(x)
        if self.act_split:
            xq = xh + h(x - xh)
        elif self.ext is not None and name.endswith(LN_FED):
            thr, cap = self.ext
            amax = x.reshape(-1, x.shape[-1]).abs().amax(dim=0)
            med = amax.median()
            idx = torch.nonzero(amax > thr * med).flatten()
            if idx.numel() > cap:
                idx = idx[amax[idx].argsort(descending=True)[:cap]]
            self.nflag[name] = max(self.nflag.get(name, 0), int(idx.numel()))
            xq = xh.clone()
            xq[..., idx] = xh[..., idx] + h(x[..., idx] - xh[..., idx])
        else:
            xq = xh
        y = xq @ wq.t()
        return y + b if b is not None else y

    @staticmethod
    def attention(q, k, v, opq=None, sdpa=False):
        sc = q.shape[-1] ** -0.5
        s = (h(q * (sc * math.log2(math.e))) @ h(k).transpose(-2, -1))
        p = torch.exp2(s - s.amax(dim=-1, keepdim=True))
        pr = h(p)
        return h((pr @ h(v)) / pr.sum(dim=-1, keepdim=True))


def run_emu(emu, sde, sdd, cfg, imgs, ts, mb=None):
    ol, oa = R.linear, R.softmax_attention
    R.linear, R.softmax_attention = emu.linear, emu.attention
    try:
        with torch.no_grad():
            return R.run_scene(sde, sdd, cfg, imgs, ts, mem_batches=mb, sdpa=False)
    finally:
        R.linear, R.softmax_attention = ol, oa


rel = lambda a, b: float((a - b).abs().max() / b.abs().max())


def massive():
    cfg = SMALL
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    sde = {k: v.clone() for k, v in sde.items()}; sdd = {k: v.clone() for k, v in sdd.items()}
    sde["blocks_enc.0.mlp.fc2.weight"][[5, 77]] *= 4.0e3
    sdd["blocks_dec.0.mlp.fc2.weight"][[3, 90]] *= 4.0e3
    imgs, ts = S.make_images(3, 224, 224, 2)
    with torch.no_grad():
        u0, r0, _ = R.run_scene(sde, sdd, cfg, imgs, ts, mem_batches=[2, 1], sdpa=False)
    for label, kw in (("fp16w2", {}), ("fp16w2 + full activation split (3 passes)", dict(act_split=True)),
                      ("fp16w2 + outlier ext thr 4 cap 64", dict(ext=(4.0, 64))), ("fp16w2 + outlier ext thr 4 cap 8", dict(ext=(4.0, 8))),
                      ("fp16w2 + outlier ext thr 8 cap 32", dict(ext=(8.0, 32))), ("fp16 + outlier ext thr 4 cap 64", dict(ext=(4.0, 64), split=lambda n: False))):
        emu = Emu((("e.", sde), ("d.", sdd)), **kw)
        u, r, _ = run_emu(emu, sde, sdd, cfg, imgs, ts, [2, 1])
        print(f"{label:46s} update {rel(u, u0):.3e} render {rel(r, r0):.3e}  max flagged {max(emu.nflag.values()) if emu.nflag else 0}", flush=True)


def split(V=3):
    cfg = MUST3R_224
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(V, 224, 224, 0)
    with torch.no_grad():
        u0, r0, _ = R.run_scene(sde, sdd, cfg, imgs, ts, sdpa=False)
    sets = {
        "all split (fp16w2)": lambda n: True,
        "none split (fp16)": lambda n: False,
        "encoder only split": lambda n: n.startswith("e."),
        "decoder only split": lambda n: n.startswith("d."),
        "all but decoder block GEMMs of update+render (dec blocks plain)": lambda n: not n.startswith("d.blocks_dec"),
        "all but decoder MLPs": lambda n: not (n.startswith("d.blocks_dec") and ".mlp." in n),
        "all but decoder fc2 + proj + cross proj (K-long / N=768)": lambda n: not (n.startswith("d.blocks_dec") and (n.endswith("fc2.weight") or n.endswith("proj.weight"))),
        "all but MLPs (enc+dec)": lambda n: ".mlp." not in n,
        "all but attention projections (qkv/proj/projq/k/v)": lambda n: ".mlp." in n or "blocks" not in n,
        "only fc1/fc2 of the encoder split": lambda n: n.startswith("e.") and ".mlp." in n,
        "all but encoder MLPs": lambda n: not (n.startswith("e.") and ".mlp." in n),
        "all but MLPs and feedback MLP": lambda n: ".mlp." not in n and "feedback" not in n,
        "all but fc1 (enc+dec)": lambda n: "fc1" not in n,
        "all but fc2 (enc+dec)": lambda n: "fc2" not in n,
        "plain: enc fc2 + dec MLP": lambda n: not ((n.startswith("e.") and "fc2" in n) or (n.startswith("d.") and (".mlp." in n or "feedback" in n))),
        "plain: enc fc1 + dec MLP": lambda n: not ((n.startswith("e.") and "fc1" in n) or (n.startswith("d.") and (".mlp." in n or "feedback" in n))),
        "plain: enc MLP + dec fc2": lambda n: not ((n.startswith("e.") and ".mlp." in n) or (n.startswith("d.") and "fc2" in n)),
        "plain: enc MLP of blocks 12-23 + dec MLP": lambda n: not ((n.startswith("e.") and ".mlp." in n and int(n.split(".")[2]) >= 12) or (n.startswith("d.") and (".mlp." in n or "feedback" in n))),
        "plain: enc MLP of blocks 0-11 + dec MLP": lambda n: not ((n.startswith("e.") and ".mlp." in n and int(n.split(".")[2]) < 12) or (n.startswith("d.") and (".mlp." in n or "feedback" in n))),
    }
    att = lambda n: ".mlp." not in n   # the fp16wa set: everything but the Mlp Linears
    sets.update({
        "wa: all but MLPs (= fp16wa)": att,
        "wa minus encoder qkv": lambda n: att(n) and not (n.startswith("e.") and "attn.qkv" in n),
        "wa minus encoder proj": lambda n: att(n) and not (n.startswith("e.") and "attn.proj" in n),
        "wa minus encoder attention (qkv + proj)": lambda n: att(n) and not (n.startswith("e.") and ".attn." in n),
        "wa minus decoder self-attention qkv": lambda n: att(n) and not (n.startswith("d.") and ".attn.qkv" in n),
        "wa minus decoder cross-attention projections": lambda n: att(n) and not (n.startswith("d.") and "cross_attn" in n),
        "wa minus all proj (enc + dec output projections)": lambda n: att(n) and not n.endswith("proj.weight"),
        "wa minus encoder attention of blocks 12-23": lambda n: att(n) and not (n.startswith("e.") and ".attn." in n and int(n.split(".")[2]) >= 12),
        "wa minus encoder attention of blocks 0-11": lambda n: att(n) and not (n.startswith("e.") and ".attn." in n and int(n.split(".")[2]) < 12),
    })
    if len(sys.argv) > 3:
        sets = {k: v for k, v in sets.items() if any(o in k for o in sys.argv[3:])}
    for label, fn in sets.items():
        emu = Emu((("e.", sde), ("d.", sdd)), split=fn)
        u, r, _ = run_emu(emu, sde, sdd, cfg, imgs, ts)
        print(f"V={V} {label:64s} update {rel(u, u0):.3e} render {rel(r, r0):.3e}", flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    if sys.argv[1] == "parts":
        pass
    elif sys.argv[1] == "massive":
        massive()
    else:
        split(int(sys.argv[2]) if len(sys.argv) > 2 else 3)


def massive_parts():
    """which rounding carries the massive-activation error?  one class of operands rounded at a time"""
    cfg = SMALL
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    sde = {k: v.clone() for k, v in sde.items()}; sdd = {k: v.clone() for k, v in sdd.items()}
    sde["blocks_enc.0.mlp.fc2.weight"][[5, 77]] *= 4.0e3
    sdd["blocks_dec.0.mlp.fc2.weight"][[3, 90]] *= 4.0e3
    imgs, ts = S.make_images(3, 224, 224, 2)
    with torch.no_grad():
        u0, r0, _ = R.run_scene(sde, sdd, cfg, imgs, ts, mem_batches=[2, 1], sdpa=False)

    class Part(Emu):
        def __init__(self, sds, which):
            super().__init__(sds)
            self.which = which

        def linear(self, x, w, b, opq=None):
            name = self.names.get(id(w), "?")
            if name.endswith("head_dec.proj.weight") or name == "?":
                y = x @ w.t()
                return y + b if b is not None else y
            ln_fed = name.endswith(LN_FED)
            xq = h(x) if (("act_ln" in self.which and ln_fed) or ("act_other" in self.which and not ln_fed)) else x
            y = xq @ w.t()
            y = y + b if b is not None else y
            if "out_qkv" in self.which and (name.endswith("attn.qkv.weight") or "cross_attn.projq" in name or "cross_attn.projk" in name or "cross_attn.projv" in name):
                y = h(y)
            return y

        def attention(self, q, k, v, opq=None, sdpa=False):
            if "attn" in self.which:
                return Emu.attention(q, k, v)
            sc = q.shape[-1] ** -0.5
            return torch.softmax((q @ k.transpose(-2, -1)) * sc, dim=-1) @ v

    for which in (("act_ln",), ("act_other",), ("out_qkv",), ("attn",), ("act_ln", "act_other", "attn")):
        emu = Part((("e.", sde), ("d.", sdd)), which)
        u, r, _ = run_emu(emu, sde, sdd, cfg, imgs, ts, [2, 1])
        print(f"{'+'.join(which):30s} update {rel(u, u0):.3e} render {rel(r, r0):.3e}", flush=True)


if __name__ == "__main__" and sys.argv[1] == "parts":
    massive_parts()
