"""GPU (-m gpu): edge cases of the forward path against the CPU oracle.

* RoPE F0 != 1: a checkpoint run at a non-native image size (``load_model(img_size=...)`` rewrites ``pos_embed`` to
  ``RoPE100_<old>:<new>``, model/__init__.py:97-107 -> blocks/pos_embed.py:12-19) -- encoder AND decoder tables.
* ``ManyAR_PatchEmbed``: portrait views stored landscape and transposed before the patch projection (dust3r leaf,
  SURVEY.md Appendix A), alone and mixed with landscape views in one batch.
* fp16 operand range: trained ViTs carry "massive activations" (a few residual channels 1e3-1e4 times the rest) -- every
  16-bit operand of the path is downstream of a LayerNorm or bounded, so the fp16 modes must keep their parity with such
  channels injected; and an overflow of the fp16 range (|v| > 65504, here a 3e4 x inflated MLP) must saturate, never produce
  inf/NaN, with the bf16 mode as the documented fallback.
"""
import dataclasses

import pytest
import torch

from must3r_amd import synthetic as S
from must3r_amd.config import TINY, SMALL
from util import TOL, rel_inf
from test_ops_gpu import record

pytestmark = pytest.mark.gpu


def _modules(cfg, sde, sdd, precision="fp16w2", pos_embed=None, patch_embed=None):
    import must3r_amd.model as M
    kw = {} if pos_embed is None else {"pos_embed": pos_embed}
    ekw = dict(kw) if patch_embed is None else dict(kw, patch_embed=patch_embed)
    enc = M.Dust3rEncoder(img_size=(cfg.img_size,) * 2, embed_dim=cfg.enc_dim, depth=cfg.enc_depth, num_heads=cfg.enc_heads,
                          precision=precision, **ekw)
    dec = M.MUSt3R(img_size=(cfg.img_size,) * 2, enc_embed_dim=cfg.enc_dim, embed_dim=cfg.dec_dim, depth=cfg.dec_depth,
                   num_heads=cfg.dec_heads, feedback_type="single_mlp", memory_mode="kv", landscape_only=False,
                   precision=precision, **kw)
    enc.load_state_dict(sde, strict=True)
    dec.load_state_dict(sdd, strict=True)
    return enc.cuda().eval(), dec.cuda().eval()


def _scene_errs(enc, dec, sde, sdd, cfg, imgs, ts, mb):
    from must3r_amd.engine import run_scene
    from oracle import must3r_ref as R
    out = run_scene(enc, dec, imgs.cuda(), ts.cuda(), mem_batches=mb)
    torch.cuda.synchronize()
    with torch.no_grad():
        xo, _ = R.encoder_forward(sde, cfg, imgs, ts)
        updo, reno, _ = R.run_scene(sde, sdd, cfg, imgs, ts, mem_batches=mb)
    return out, dict(x=rel_inf(out["x"].cpu(), xo), update=rel_inf(out["update"].cpu(), updo), render=rel_inf(out["render"].cpu(), reno))


@pytest.mark.parametrize("precision", ["fp16w2", "bf16"])
def test_rope_f0_non_native_image_size(precision):
    """A 64-pixel model run at 96 pixels: pos_embed 'RoPE100_64:96' -> F0 = 64/96 in BOTH modules' RoPE tables."""
    cfg = dataclasses.replace(TINY, img_size=96, rope_f0=64.0 / 96.0)
    sde, sdd = S.make_encoder_state_dict(cfg, 3), S.make_decoder_state_dict(cfg, 3)
    enc, dec = _modules(cfg, sde, sdd, precision, pos_embed="RoPE100_64:96")
    assert abs(enc.cfg.rope_f0 - 64.0 / 96.0) < 1e-7 and abs(dec.cfg.rope_f0 - 64.0 / 96.0) < 1e-7
    imgs, ts = S.make_images(3, 80, 96, 21)
    _, errs = _scene_errs(enc, dec, sde, sdd, cfg, imgs, ts, [2, 1])
    # and the F0 = 1 tables would NOT match: the test really exercises the rescaled frequencies
    enc1, dec1 = _modules(cfg, sde, sdd, precision)
    _, errs1 = _scene_errs(enc1, dec1, sde, sdd, cfg, imgs, ts, [2, 1])
    record("rope_f0", precision=precision, **errs, render_with_f0_1=errs1["render"])
    assert max(errs.values()) < TOL[precision], errs
    # (the wrong tables cost ~6e-2 in every precision: far outside the tolerance and the measured error)
    assert errs1["render"] > 4 * TOL[precision] and errs1["render"] > 5 * max(errs.values()), errs1


@pytest.mark.parametrize("layout", ["portrait", "mixed"])
def test_many_ar_patch_embed_portrait(layout):
    """patch_embed='ManyAR_PatchEmbed': the batch is stored landscape [B,3,48,64]; a portrait view (true_shape 64x48) is
    transposed before the projection and gets the positions of the transposed grid."""
    from oracle import must3r_ref as R
    cfg = TINY
    sde, sdd = S.make_encoder_state_dict(cfg, 5), S.make_decoder_state_dict(cfg, 5)
    enc, _ = _modules(cfg, sde, sdd, "fp16w2", patch_embed="ManyAR_PatchEmbed")
    imgs, _ = S.make_images(3, 48, 64, 9)
    portrait = [True, True, True] if layout == "portrait" else [False, True, False]
    ts = torch.tensor([[64, 48] if p else [48, 64] for p in portrait], dtype=torch.int64)
    x, pos = enc(imgs.cuda(), ts.cuda())
    torch.cuda.synchronize()
    errs = []
    for v, p in enumerate(portrait):
        im = imgs[v:v + 1].swapaxes(-1, -2) if p else imgs[v:v + 1]
        with torch.no_grad():
            xo, po = R.encoder_forward(sde, cfg, im.contiguous())
        assert torch.equal(pos[v].cpu(), po[0]), f"positions of view {v}"
        errs.append(rel_inf(x[v].cpu(), xo[0]))
    record("many_ar", layout=layout, errs=errs)
    assert max(errs) < TOL["fp16w2"], errs
    if layout == "mixed":   # the portrait view really went through the transposed path
        with torch.no_grad():
            x_plain, _ = R.encoder_forward(sde, cfg, imgs[1:2])
        assert rel_inf(x[1].cpu(), x_plain[0]) > 0.1


def _inflate(sd, key, rows, factor):
    sd = {k: v.clone() for k, v in sd.items()}
    sd[key][rows] *= factor
    return sd


@pytest.mark.parametrize("precision", ["fp16w2", "fp16"])
def test_massive_activations_keep_fp16_parity(precision):
    """Two residual channels of the encoder and of the decoder carry ~1e3-1e4 x the magnitude of the others from the first
    block on (the MLP output rows that feed them are inflated).  RANGE: the residual stream is fp32 and every 16-bit operand
    is behind a LayerNorm or a bounded epilogue, so nothing overflows (finite outputs, no saturation).  PRECISION: after the
    LayerNorm the two channels are O(sqrt(C/2)) and all others O(1e-3); the 2^-11 rounding of the big LN outputs is then as
    large as ~10 % of what the small channels contribute to a GEMM output, so ANY 11-bit activation format loses the 1e-3
    parity on such weights (measured on gfx950: 1.0-1.6e-2; splitting the weights does not help, the activations are the
    term).  The reference runs these GEMMs under bf16 autocast (2^-8): the same scene in the bf16 operand mode is several
    times further away, which is what the test pins -- fp16 stays the better format, with a stated bound."""
    from oracle import must3r_ref as R
    cfg = SMALL
    sde = _inflate(S.make_encoder_state_dict(cfg, 0), "blocks_enc.0.mlp.fc2.weight", [5, 77], 4.0e3)
    sdd = _inflate(S.make_decoder_state_dict(cfg, 0), "blocks_dec.0.mlp.fc2.weight", [3, 90], 4.0e3)
    imgs, ts = S.make_images(3, 224, 224, 2)
    with torch.no_grad():   # size of the injected channels in the encoder's residual stream after block 0
        x0, p0 = R.patch_embed(sde, imgs[:1], cfg.patch_size)
        h = R.layer_norm(x0, sde["blocks_enc.0.norm1.weight"], sde["blocks_enc.0.norm1.bias"], 1e-6)
        x1 = x0 + R.self_attention(sde, "blocks_enc.0.attn", h, p0, cfg.enc_heads, cfg)
        x1 = x1 + R.mlp(sde, "blocks_enc.0.mlp", R.layer_norm(x1, sde["blocks_enc.0.norm2.weight"], sde["blocks_enc.0.norm2.bias"], 1e-6))
    big = float(x1[..., [5, 77]].abs().max())
    rest = float(x1[..., [c for c in range(cfg.enc_dim) if c not in (5, 77)]].abs().median())
    enc, dec = _modules(cfg, sde, sdd, precision)
    out, errs = _scene_errs(enc, dec, sde, sdd, cfg, imgs, ts, [2, 1])
    record("massive_activations", precision=precision, channel_absmax=big, other_abs_median=rest, **errs)
    assert big > 1.0e3 and big / rest > 1.0e3, (big, rest)
    assert torch.isfinite(out["render"]).all() and torch.isfinite(out["update"]).all() and torch.isfinite(out["x"]).all()
    encb, decb = _modules(cfg, sde, sdd, "bf16")
    _, errs_b = _scene_errs(encb, decb, sde, sdd, cfg, imgs, ts, [2, 1])
    record("massive_activations_bf16", precision=precision, **errs_b)
    assert max(errs.values()) < 3.0e-2, errs                       # stated bound of the fp16 modes on massive activations
    assert max(errs.values()) < 0.5 * max(errs_b.values()), (errs, errs_b)   # and well inside what 8-bit mantissas give


@pytest.mark.parametrize("precision", ["fp16wa", "fp16w2", "fp16"])
def test_token_constant_massive_activations_stay_inside_the_target(precision):
    """The massive-activation pattern reported for trained ViTs: a few residual channels carry a huge value that is (nearly) the SAME for
    every token (here: 3e3 from a bias of block 0's MLP output, encoder and decoder, ~1e3 x the other channels).  Unlike the adversarial
    case above (token-DEPENDENT outliers, 1.0-1.6e-2) the 16-bit operands keep the 1e-3 parity: the outlier channels are constant
    after the LayerNorm too, their 2^-11 rounding is a constant offset of every GEMM output (emulation: 3e-6, scripts/emul/gemm_precision.py
    model with a constant bias).  Exercises the LN-fold path of the one-view update as well (its fp16 copy of the raw residual rows holds
    the outliers at an ulp of 2)."""
    from oracle import must3r_ref as R
    cfg = SMALL
    sde = {k: v.clone() for k, v in S.make_encoder_state_dict(cfg, 0).items()}
    sdd = {k: v.clone() for k, v in S.make_decoder_state_dict(cfg, 0).items()}
    sde["blocks_enc.0.mlp.fc2.bias"][[5, 77]] = 3.0e3
    sdd["blocks_dec.0.mlp.fc2.bias"][[3, 90]] = 3.0e3
    imgs, ts = S.make_images(3, 224, 224, 2)
    with torch.no_grad():
        x0, p0 = R.patch_embed(sde, imgs[:1], cfg.patch_size)
        h = R.layer_norm(x0, sde["blocks_enc.0.norm1.weight"], sde["blocks_enc.0.norm1.bias"], 1e-6)
        x1 = x0 + R.self_attention(sde, "blocks_enc.0.attn", h, p0, cfg.enc_heads, cfg)
        x1 = x1 + R.mlp(sde, "blocks_enc.0.mlp", R.layer_norm(x1, sde["blocks_enc.0.norm2.weight"], sde["blocks_enc.0.norm2.bias"], 1e-6))
    big = x1[..., [5, 77]]
    rest = float(x1[..., [c for c in range(cfg.enc_dim) if c not in (5, 77)]].abs().median())
    assert float(big.abs().min()) > 2.0e3 and float(big.std()) < 0.01 * float(big.abs().mean()) and float(big.abs().mean()) / rest > 1.0e3
    enc, dec = _modules(cfg, sde, sdd, precision)
    out, errs = _scene_errs(enc, dec, sde, sdd, cfg, imgs, ts, [2, 1])
    record("massive_activations_token_constant", precision=precision, channel_abs=float(big.abs().mean()), other_abs_median=rest, **errs)
    assert torch.isfinite(out["render"]).all() and torch.isfinite(out["update"]).all()
    assert max(errs.values()) < 1.0e-3, errs


def test_fp16_overflow_saturates_and_bf16_is_the_fallback():
    """An MLP whose hidden activations exceed the fp16 range (fc1 inflated 3e4 x): the fp16 epilogue stores saturate at
    +-65504 -- the outputs stay finite (no inf -> NaN chain through LayerNorm / softmax) -- and the bf16 operand mode, whose
    range is fp32's, stays inside its own tolerance on the same weights."""
    cfg = SMALL
    sde = S.make_encoder_state_dict(cfg, 0)
    sde = {k: v.clone() for k, v in sde.items()}
    sde["blocks_enc.1.mlp.fc1.weight"] *= 3.0e4
    sdd = S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(2, 224, 224, 4)
    enc, dec = _modules(cfg, sde, sdd, "fp16w2")
    out16, errs16 = _scene_errs(enc, dec, sde, sdd, cfg, imgs, ts, [2])
    enc.precision = dec.precision = "bf16"
    outb, errsb = _scene_errs(enc, dec, sde, sdd, cfg, imgs, ts, [2])
    record("fp16_overflow", fp16w2=errs16, bf16=errsb)
    assert torch.isfinite(out16["x"]).all() and torch.isfinite(out16["render"]).all() and torch.isfinite(out16["update"]).all()
    assert max(errsb.values()) < TOL["bf16"], errsb
