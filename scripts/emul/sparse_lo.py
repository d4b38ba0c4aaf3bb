"""CPU emulation of a 2:4-SPARSE low part for the split-weight GEMMs (experiment infrastructure, never imported by the product).

  python scripts/emul/sparse_lo.py [views]

Model: every split Linear = fp32 accumulation of  x16 . W_hi^T  +  x16 . P(W_lo)^T  with W_hi = fp16(W), W_lo = fp16(W - W_hi) and P() keeping, in every
group of 4 consecutive k of a row, the 2 entries of largest magnitude (the structure v_smfmac_f32_16x16x64_f16 multiplies at twice the dense rate with the
sparse matrix as its A operand = our weight tile; the activations stay dense fp16).  The dropped half of W_lo is the SMALL half: ~20 % of the residual's
energy for a uniform rounding error, so the weight-rounding error of a split Linear falls to ~45 % of the unsplit one instead of to ~0.
The question: does `fp16wa` with sparse low parts stay inside the budget (fp16wa dense: 5.6e-4 in this emulation) -- before any kernel is written.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import must3r_ref as R
from must3r_amd import synthetic as S
from must3r_amd.config import MUST3R_224
from scripts.emul.gemm_precision import Emu, run_emu, rel, h


def prune24(w):
    """keep the 2 largest |.| of every 4 consecutive entries of the last dim"""
    g = w.reshape(*w.shape[:-1], w.shape[-1] // 4, 4)
    idx = g.abs().argsort(dim=-1, descending=True)[..., :2]
    mask = torch.zeros_like(g, dtype=torch.bool).scatter_(-1, idx, True)
    return (g * mask).reshape(w.shape)


class EmuSparse(Emu):
    def __init__(self, sds, mode):
        super().__init__(sds)
        self.mode, self.cache = mode, {}

    def linear(self, x, w, b, opq=None):
        name = self.names.get(id(w), "?")
        if name.endswith("head_dec.proj.weight") or name == "?":
            y = x @ w.t()
            return y + b if b is not None else y
        m = self.mode(name)
        wh, xh = h(w), h(x)
        y = xh @ wh.t()
        if m is not None:
            if id(w) not in self.cache:
                lo = h(w - wh)
                self.cache[id(w)] = prune24(lo) if m == "sparse" else lo
            y = y + xh @ self.cache[id(w)].t()
        return y + b if b is not None else y


def main(V):
    cfg = MUST3R_224
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(V, 224, 224, 0)
    with torch.no_grad():
        u0, r0, _ = R.run_scene(sde, sdd, cfg, imgs, ts, sdpa=False)
    att = lambda n: ".mlp." not in n  # noqa: E731
    sets = {
        "none (fp16)": lambda n: None,
        "wa: attention-side dense split, Mlp plain": lambda n: "split" if att(n) else None,
        "wa with 2:4-sparse low parts": lambda n: "sparse" if att(n) else None,
        "all split (fp16w2)": lambda n: "split",
        "all 2:4-sparse low parts": lambda n: "sparse",
        "attention-side dense split, Mlp 2:4-sparse low parts": lambda n: "split" if att(n) else "sparse",
    }
    for label, fn in sets.items():
        emu = EmuSparse((("e.", sde), ("d.", sdd)), fn)
        u, r, _ = run_emu(emu, sde, sdd, cfg, imgs, ts)
        print(f"V={V} {label:60s} update {rel(u, u0):.3e} render {rel(r, r0):.3e}", flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
