"""``Dust3rEncoder`` -- drop-in for must3r/model/encoder.py:13 backed by libmust3r_hip.

Same constructor arguments, attributes (``patch_size``, ``embed_dim``, ``depth``, ``grid_size``,
``patch_embed``), state-dict keys and ``forward(img, true_shape) -> (x, pos)`` contract
(encoder.py:46-52).  The forward is ONE native call (``must3r_hip_encode``): patch im2col + MFMA GEMM,
24 x [LN -> QKV GEMM with fused 2-D RoPE -> flash self-attention -> proj GEMM with fused residual ->
LN -> fc1 GEMM with fused GELU -> fc2 GEMM with fused residual], final LN.

Precision: the reference disables autocast here (encoder.py:46) and runs fp32 (TF32 on NVIDIA,
demo.py:12).  gfx950 has no TF32; this module feeds the MFMAs with ``precision`` operands
('fp16wa' default: fp16, split weights in the attention-side Linears, plain in the Mlp Linears; 'fp16w2': every weight split;
'fp16': 10-bit mantissa like TF32; 'bf16') and keeps the residual stream, LayerNorm, softmax
and accumulators in fp32.  Autocast state is ignored, as in the reference.
"""
from functools import partial

import torch
import torch.nn as nn

from .. import _lib
from ..config import ModelConfig
from .blocks import EncBlockParams, PatchEmbedParams, init_weights, parse_pos_embed
from ._hip_module import HipModule, operand_dtype


class Dust3rEncoder(HipModule):
    _part = _lib.PART_ENCODER
    _prefix = "encoder."

    def __init__(self, img_size=(224, 224), patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), patch_embed="PatchEmbedDust3R", pos_embed="RoPE100",
                 precision="fp16wa", **kv):
        super().__init__()
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        self.embed_dim = embed_dim
        self.depth = depth
        self.patch_size = patch_size
        self.patch_embed = PatchEmbedParams(patch_embed, img_size, patch_size, embed_dim)
        self.max_seq_len = max(img_size) // patch_size
        self.grid_size = self.patch_embed.grid_size
        freq, f0 = parse_pos_embed(pos_embed)
        self.blocks_enc = nn.ModuleList([EncBlockParams(embed_dim, mlp_ratio) for _ in range(depth)])
        self.norm_enc = nn.LayerNorm(embed_dim, eps=1e-6)
        init_weights(self)
        self._hip_init(ModelConfig(img_size=max(img_size), patch_size=patch_size, enc_dim=embed_dim, enc_depth=depth,
                                   enc_heads=num_heads, mlp_ratio=mlp_ratio, rope_freq=freq, rope_f0=f0), precision)

    @torch.no_grad()
    def forward(self, img, true_shape=None):
        ctx = self._context()
        dev = self._ctx_dev
        img = self._check_input(img, "img", torch.float32)
        B, Cin, H, W = img.shape
        assert Cin == 3 and H % self.patch_size == 0 and W % self.patch_size == 0, (Cin, H, W)
        if self.patch_embed.kind == "ManyAR_PatchEmbed" and true_shape is not None:
            # dust3r ManyAR_PatchEmbed: batch stored landscape, portrait samples
            # (true_shape h > w) are transposed before the patch projection.
            ts = true_shape.to("cpu")
            portrait = ts[:, 0] > ts[:, 1]
            if bool(portrait.any()) and not bool(portrait.all()):
                x = torch.empty((B, (H // 16) * (W // 16), self.embed_dim), dtype=torch.float32, device=img.device)
                pos = torch.empty((B, x.shape[1], 2), dtype=torch.int64, device=img.device)
                idx_p = portrait.nonzero().flatten().to(img.device)
                idx_l = (~portrait).nonzero().flatten().to(img.device)
                x[idx_l], pos[idx_l] = self._encode(ctx, dev, img[idx_l].contiguous())
                x[idx_p], pos[idx_p] = self._encode(ctx, dev, img[idx_p].swapaxes(-1, -2).contiguous())
                return x, pos
            if bool(portrait.all()):
                img = img.swapaxes(-1, -2).contiguous()
        return self._encode(ctx, dev, img)

    def _encode(self, ctx, dev, img):
        B, _, H, W = img.shape
        N = (H // 16) * (W // 16)
        x = torch.empty((B, N, self.embed_dim), dtype=torch.float32, device=img.device)
        pos = torch.empty((B, N, 2), dtype=torch.int64, device=img.device)
        code = operand_dtype(self.precision) | (_lib.ATTN_FP8 if self.attention_fp8 else 0)
        _lib.check(ctx.lib.must3r_hip_encode(ctx.handle, code, img.data_ptr(), B, H, W,
                                             x.data_ptr(), pos.data_ptr(), self._stream(dev)))
        return x, pos

    def from_dust3r(self, state_dict, verbose=True):  # encoder.py:54-61
        state_dict = {k.replace("enc_blocks", "blocks_enc").replace("enc_norm", "norm_enc"): v
                      for k, v in state_dict.items()}
        inc = self.load_state_dict(state_dict, strict=False)
        if verbose:
            print(inc)
        assert len(inc.missing_keys) == 0
        return inc

    def from_croco(self, state_dict, verbose=True):
        return self.from_dust3r(state_dict, verbose=verbose)
