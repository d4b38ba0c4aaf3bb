"""Parameter containers with the reference's attribute tree (hence its state-dict keys, SURVEY.md
section 8b).  They hold weights only: all arithmetic of the forward path runs in libmust3r_hip.

Reference classes mirrored (must3r/model/blocks/): ``Attention`` attention.py:82, ``CachedCrossAttention``
attention.py:129, ``Block`` layers.py:36, ``CachedDecoderBlock`` layers.py:57, ``LinearHead`` head.py:63,
croco ``Mlp`` (fc1/fc2), dust3r ``PatchEmbedDust3R`` (proj).  Initialisation follows
``BaseTransformer._init_weights`` layers.py:23-33.
"""
from enum import Enum

import torch
import torch.nn as nn

MEMORY_MODES = ["norm_y", "kv", "raw"]  # layers.py:9


class ActivationType(Enum):  # blocks/head.py:8-10
    NORM_EXP = "norm_exp"
    LINEAR = "linear"


def parse_pos_embed(name):
    """'RoPE100' / 'RoPE100_224:512' -> (freq, F0)   (blocks/pos_embed.py:7-22)."""
    assert name.startswith("RoPE"), name
    f0 = 1.0
    if "_" in name:
        name, res = name.split("_")
        old, new = res.split(":")
        f0 = float(old) / float(new)
    return float(name[len("RoPE"):]), f0


class AttnParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)


class CrossAttnParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.projq = nn.Linear(dim, dim, bias=True)
        self.projk = nn.Linear(dim, dim, bias=True)
        self.projv = nn.Linear(dim, dim, bias=True)
        self.proj = nn.Linear(dim, dim)


class MlpParams(nn.Module):
    def __init__(self, dim, hidden, out=None):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, out or dim)


class EncBlockParams(nn.Module):
    def __init__(self, dim, mlp_ratio, eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = AttnParams(dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = MlpParams(dim, int(dim * mlp_ratio))


class DecBlockParams(nn.Module):
    def __init__(self, dim, mlp_ratio, memory_mode, eps=1e-6):
        super().__init__()
        assert memory_mode in MEMORY_MODES
        self.memory_mode = memory_mode
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = AttnParams(dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.norm_y = nn.LayerNorm(dim, eps=eps)
        self.cross_attn = CrossAttnParams(dim)
        self.norm3 = nn.LayerNorm(dim, eps=eps)
        self.mlp = MlpParams(dim, int(dim * mlp_ratio))


class PatchEmbedParams(nn.Module):
    def __init__(self, name, img_size, patch_size, dim):
        super().__init__()
        assert name in ("PatchEmbedDust3R", "ManyAR_PatchEmbed"), name
        self.kind = name
        self.patch_size = (patch_size, patch_size)
        self.img_size = tuple(img_size)
        self.grid_size = (img_size[0] // patch_size, img_size[1] // patch_size)
        self.proj = nn.Conv2d(3, dim, kernel_size=patch_size, stride=patch_size)


class LinearHeadParams(nn.Module):
    def __init__(self, dim, out_dim, patch_size):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Linear(dim, out_dim, bias=True)


def init_weights(module):
    """BaseTransformer.initialize_weights (layers.py:17-33): xavier-uniform Linear, zero bias, LN 1/0."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
