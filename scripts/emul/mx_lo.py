"""CPU emulation of the "1.25-pass" split-weight GEMM (experiment infrastructure, never imported by the product).

  python scripts/emul/mx_lo.py [views] [filter ...]

Model: every Linear = fp32 accumulation of  x16 . W_hi^T  +  q(x) . q(W_lo)^T  where W_hi = fp16(W), W_lo = W - W_hi (fp32 residual),
and q() is an MX block format (32 consecutive k share one power-of-two scale): e2m1 ("fp4"), e2m3 ("fp6"), e4m3 ("fp8").
Everything else as scripts/emul/gemm_precision.py (activations / attention operands fp16, residual stream / LN / softmax fp32, head fp32).
The question: which low-part format, on which Linears, keeps the scene inside the 8e-4 budget -- before any kernel is written.
"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import must3r_ref as R
from must3r_amd import synthetic as S
from must3r_amd.config import MUST3R_224
from scripts.emul.gemm_precision import Emu, run_emu, rel, h

GRID = {
    "fp4": torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6]),                                     # e2m1
    "fp6": torch.tensor([0, .125, .25, .375, .5, .625, .75, .875, 1, 1.125, 1.25, 1.375, 1.5, 1.625, 1.75, 1.875,
                         2, 2.25, 2.5, 2.75, 3, 3.25, 3.5, 3.75, 4, 4.5, 5, 5.5, 6, 6.5, 7, 7.5]),  # e2m3
}


def q_grid(v, grid):
    """round |v| to the nearest grid point (ties: to the even index, like RNE on the code), keep the sign"""
    a = v.abs().clamp(max=float(grid[-1]))
    idx = torch.bucketize(a, (grid[1:] + grid[:-1]) / 2)   # midpoints -> nearest; exact midpoints go up (fine for an error model)
    return torch.sign(v) * grid[idx]


def mx_quant(t, fmt, unit_scale=False, blk=32):
    """MX block quantisation along the last dim, blocks of 32, scale 2^e with the block max mapped into (max/2, max]"""
    if blk != 32:   # one scale per ROW (it factors out of the K sum: no MX scale hardware needed, separate low-part accumulators instead)
        grid = GRID[fmt]
        amax = t.abs().amax(dim=-1, keepdim=True).clamp(min=1e-30)
        s = torch.exp2(torch.ceil(torch.log2(amax / float(grid[-1]))))
        return q_grid(t / s, grid) * s
    if fmt == "fp8":
        K = t.shape[-1]
        b = t.reshape(*t.shape[:-1], K // 32, 32)
        amax = b.abs().amax(dim=-1, keepdim=True).clamp(min=1e-30)
        e = torch.ceil(torch.log2(amax / 448.0)) if not unit_scale else torch.zeros_like(amax)
        s = torch.exp2(e)
        return ((b / s).to(torch.float8_e4m3fn).float() * s).reshape(t.shape)
    grid = GRID[fmt]
    K = t.shape[-1]
    b = t.reshape(*t.shape[:-1], K // 32, 32)
    amax = b.abs().amax(dim=-1, keepdim=True).clamp(min=1e-30)
    e = torch.ceil(torch.log2(amax / float(grid[-1]))) if not unit_scale else torch.zeros_like(amax)
    s = torch.exp2(e)
    return (q_grid(b / s, grid) * s).reshape(t.shape)


class EmuLo(Emu):
    """lo(name) -> None (plain fp16 weight), "split" (fp16 low part, 2 passes) or an MX format for the low-part product"""
    def __init__(self, sds, lo, x_unit_scale=False, w_blk=32, x_blk=32):
        super().__init__(sds)
        self.lo, self.x_unit, self.w_blk, self.x_blk = lo, x_unit_scale, w_blk, x_blk
        self.cache = {}

    def linear(self, x, w, b, opq=None):
        name = self.names.get(id(w), "?")
        if name.endswith("head_dec.proj.weight") or name == "?":
            y = x @ w.t()
            return y + b if b is not None else y
        mode = self.lo(name)
        wh = h(w)
        xh = h(x)
        y = xh @ wh.t()
        if mode == "split":
            y = y + xh @ h(w - wh).t()
        elif mode is not None:
            key = (id(w), mode)
            if key not in self.cache:
                self.cache[key] = mx_quant(w - wh, mode, blk=self.w_blk)
            y = y + mx_quant(xh, mode, self.x_unit, blk=self.x_blk) @ self.cache[key].t()
        return y + b if b is not None else y


def main(V, filt):
    cfg = MUST3R_224
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(V, 224, 224, 0)
    with torch.no_grad():
        u0, r0, _ = R.run_scene(sde, sdd, cfg, imgs, ts, sdpa=False)
    att = lambda n: ".mlp." not in n
    sets = {
        "none (fp16)": lambda n: None,
        "all split (fp16w2)": lambda n: "split",
        "wa: attention-side split, Mlp plain": lambda n: "split" if att(n) else None,
        "all fp8 lo": lambda n: "fp8",
        "all fp6 lo": lambda n: "fp6",
        "all fp4 lo": lambda n: "fp4",
        "attention-side fp4 lo, Mlp plain": lambda n: "fp4" if att(n) else None,
        "attention-side fp8 lo, Mlp plain": lambda n: "fp8" if att(n) else None,
        "attention-side split, Mlp fp4 lo": lambda n: "split" if att(n) else "fp4",
        "LN-fed fp4 lo, others plain": lambda n: "fp4" if n.endswith(("attn.qkv.weight", "mlp.fc1.weight", "cross_attn.projq.weight", "cross_attn.projk.weight", "cross_attn.projv.weight", "feedback_layer.fc1.weight", "feat_embed_enc_to_dec.weight")) else None,
        "LN-fed attention-side fp4 lo (qkv, projq, projk/v, enc->dec), others plain": lambda n: "fp4" if n.endswith(("attn.qkv.weight", "cross_attn.projq.weight", "cross_attn.projk.weight", "cross_attn.projv.weight", "feat_embed_enc_to_dec.weight")) else None,
        "all fp4 lo, x unit scale": lambda n: "fp4",
        "all fp4 lo, per-ROW scales (W and x)": lambda n: "fp4",
        "all fp4 lo, per-ROW W scale, per-block x": lambda n: "fp4",
        "all fp4 lo, per-block W, per-ROW x scale": lambda n: "fp4",
        "all fp4 lo but fc2 plain": lambda n: None if "fc2" in n else "fp4",
        "all fp4 lo but fc2 + output proj plain": lambda n: None if ("fc2" in n or n.endswith("proj.weight")) else "fp4",
        "all fp4 lo but output proj plain": lambda n: None if n.endswith("proj.weight") else "fp4",
        "all fp4 lo but encoder fc2 plain": lambda n: None if (n.startswith("e.") and "fc2" in n) else "fp4",
        "all fp4 lo but decoder Mlp plain": lambda n: None if (n.startswith("d.") and ".mlp." in n) else "fp4",
        "all fp4 lo but fc1 plain": lambda n: None if "fc1" in n else "fp4",
    }
    for label, fn in sets.items():
        if filt and not any(f in label for f in filt):
            continue
        emu = EmuLo((("e.", sde), ("d.", sdd)), fn, x_unit_scale="unit scale" in label,
                    w_blk=0 if ("per-ROW scales" in label or "per-ROW W" in label) else 32, x_blk=0 if ("per-ROW scales" in label or "per-ROW x" in label) else 32)
        u, r, _ = run_emu(emu, sde, sdd, cfg, imgs, ts)
        print(f"V={V} {label:76s} update {rel(u, u0):.3e} render {rel(r, r0):.3e}", flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3, sys.argv[2:])
