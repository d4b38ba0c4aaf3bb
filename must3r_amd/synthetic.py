"""Seeded random-init weights and inputs with the reference's state-dict contract.

No checkpoint can be fetched (no network), so tests and ``bench.py`` run on weights drawn the way the
reference initialises them (``BaseTransformer._init_weights`` must3r/model/blocks/layers.py:23-33:
xavier-uniform Linear weights, LayerNorm 1/0; ``image2_embed ~ N(0, .02)`` decoder.py:52; Conv2d
default init for ``patch_embed.proj``) with three deliberate departures so that parity tests exercise
every term: biases and LayerNorm affine parameters are perturbed instead of exactly 0/1, and
``feedback_layer.fc2`` is non-zero (the reference zero-inits it, feedback_mechanism.py:26-36, which
would make the feedback path a no-op).

The key set / shapes are exactly those of SURVEY.md section 8b so that the reference modules accept
the dicts with ``load_state_dict(strict=True)``.
"""
import math

import numpy as np
import torch

from .config import ModelConfig


def _xavier(g, out_f, in_f):
    bound = math.sqrt(6.0 / (in_f + out_f))
    return (torch.rand((out_f, in_f), generator=g) * 2 - 1) * bound


def _bias(g, n, std=0.02):
    return torch.randn((n,), generator=g) * std


def _ln(g, sd, prefix, dim):
    sd[prefix + ".weight"] = 1.0 + torch.randn((dim,), generator=g) * 0.05
    sd[prefix + ".bias"] = torch.randn((dim,), generator=g) * 0.02


def _linear(g, sd, prefix, out_f, in_f):
    sd[prefix + ".weight"] = _xavier(g, out_f, in_f)
    sd[prefix + ".bias"] = _bias(g, out_f)


def make_encoder_state_dict(cfg: ModelConfig, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    C, p = cfg.enc_dim, cfg.patch_size
    fan_in = 3 * p * p
    bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +)
    sd["patch_embed.proj.weight"] = (torch.rand((C, 3, p, p), generator=g) * 2 - 1) * bound
    sd["patch_embed.proj.bias"] = (torch.rand((C,), generator=g) * 2 - 1) * bound
    for i in range(cfg.enc_depth):
        b = f"blocks_enc.{i}"
        _ln(g, sd, b + ".norm1", C)
        _linear(g, sd, b + ".attn.qkv", 3 * C, C)
        _linear(g, sd, b + ".attn.proj", C, C)
        _ln(g, sd, b + ".norm2", C)
        _linear(g, sd, b + ".mlp.fc1", cfg.mlp_ratio * C, C)
        _linear(g, sd, b + ".mlp.fc2", C, cfg.mlp_ratio * C)
    _ln(g, sd, "norm_enc", C)
    return sd


def make_decoder_state_dict(cfg: ModelConfig, seed: int = 0, feedback_type="single_mlp"):
    g = torch.Generator().manual_seed(seed + 7919)
    sd = {}
    C, D = cfg.enc_dim, cfg.dec_dim
    sd["image2_embed"] = torch.randn((1, 1, D), generator=g) * 0.02
    _linear(g, sd, "feat_embed_enc_to_dec", D, C)
    for i in range(cfg.dec_depth):
        b = f"blocks_dec.{i}"
        _ln(g, sd, b + ".norm1", D)
        _linear(g, sd, b + ".attn.qkv", 3 * D, D)
        _linear(g, sd, b + ".attn.proj", D, D)
        _ln(g, sd, b + ".norm2", D)
        _ln(g, sd, b + ".norm_y", D)
        for n in ("projq", "projk", "projv", "proj"):
            _linear(g, sd, b + ".cross_attn." + n, D, D)
        _ln(g, sd, b + ".norm3", D)
        _linear(g, sd, b + ".mlp.fc1", cfg.mlp_ratio * D, D)
        _linear(g, sd, b + ".mlp.fc2", D, cfg.mlp_ratio * D)
    if feedback_type == "single_mlp":
        _linear(g, sd, "feedback_layer.fc1", 4 * D, D)
        sd["feedback_layer.fc2.weight"] = torch.randn((D, 4 * D), generator=g) * 0.02
        sd["feedback_layer.fc2.bias"] = _bias(g, D)
        _ln(g, sd, "feedback_norm", D)
    elif feedback_type == "single_linear":      # feedback_mechanism.py:15-17
        sd["feedback_layer.weight"] = torch.randn((D, D), generator=g) * 0.02
        sd["feedback_layer.bias"] = _bias(g, D)
        _ln(g, sd, "feedback_norm", D)
    else:
        assert not feedback_type
    _ln(g, sd, "norm_dec", D)
    _linear(g, sd, "head_dec.proj", cfg.output_dim, D)
    return sd


def make_images(n_views: int, H: int, W: int, seed: int = 0):
    """``randn(V,3,H,W)`` fp32 (what the reference's own smoke test feeds, decoder.py:580) and the
    matching ``true_shape`` int64[V,2] = (H, W)."""
    g = torch.Generator().manual_seed(seed + 104729)
    imgs = torch.randn((n_views, 3, H, W), generator=g)
    true_shape = torch.tensor([[H, W]] * n_views, dtype=torch.int64)
    return imgs, true_shape


def make_cam_pointmaps(*lead_hw, focal=40.0, noise=0.01, seed=None):
    """Raw head outputs [..., H, W, 7] whose activation is a noisy pinhole scene: local points on rays of a camera with
    the given focal, world points = a random rigid motion of them."""
    if seed is not None:
        torch.manual_seed(seed)
    *lead, H, W = lead_hw
    n = 1
    for d in lead:
        n *= d
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    z = 1.0 + torch.rand(n, H, W)
    local = torch.stack(((xs - W / 2) / focal * z, (ys - H / 2) / focal * z, z), dim=-1)
    local = local + noise * torch.randn_like(local)
    A = torch.randn(n, 3, 3)
    Q, _ = torch.linalg.qr(A)
    Q = Q * torch.sign(torch.det(Q)).view(n, 1, 1)
    t = torch.randn(n, 1, 1, 3)
    world = torch.einsum("nij,nhwj->nhwi", Q, local) + t + noise * torch.randn_like(local)

    def inv_norm_exp(p):   # raw v with v/|v| * expm1(|v|) = p
        d = p.norm(dim=-1, keepdim=True)
        return p / d.clip(min=1e-8) * torch.log1p(d)
    pm = torch.cat((inv_norm_exp(world), inv_norm_exp(local), torch.randn(n, H, W, 1)), dim=-1)
    return pm.reshape(*lead, H, W, 7)


def make_overlap_frames(seed, n_kf=3, H=24, W=32):
    """Synthetic keyframe sequence: pointmaps on a wavy surface seen from moving camera centres."""
    rng = np.random.default_rng(seed)
    frames = []
    for k in range(n_kf + 1):
        cam = np.array([0.3 * k, 0.05 * k, 0.0], dtype=np.float32)
        ys, xs = np.meshgrid(np.linspace(-1, 1, H, dtype=np.float32), np.linspace(-1.3, 1.3, W, dtype=np.float32), indexing="ij")
        z = 3.0 + 0.3 * np.sin(2 * xs + k) + 0.05 * rng.standard_normal((H, W)).astype(np.float32)
        local = np.stack((xs * z * 0.5, ys * z * 0.5, z), -1).astype(np.float32)
        pts = local + cam
        conf = (1.0 + np.exp(rng.standard_normal((H, W)))).astype(np.float32)
        frames.append(dict(pts3d=pts[None, None], pts3d_local=local[None, None], conf=conf[None, None], cam=cam))
    return frames


def make_retrieval_state_dict(dim, seed=0, prewhiten=True, postwhiten=True, hdims=None):
    """Seeded RetrievalModel state dict (reference keys, retrieval/model.py:104-151): whiteners with a non-trivial mean and a
    well-conditioned random projection (float64), xavier Linear projector with a non-zero bias.  ``hdims`` (default ``[dim]``):
    a multi-layer projector gets LayerNorms with non-trivial affine parameters between its Linears (build_projector :139-151)."""
    g = torch.Generator().manual_seed(1000 + seed)
    hdims = [dim] if hdims is None else list(hdims)
    out_dim = hdims[-1] if hdims else dim
    sd = {}
    for name, on, d in (("prewhiten", prewhiten, dim), ("postwhiten", postwhiten, out_dim)):
        if on:
            sd[name + ".m"] = (torch.randn((1, d), generator=g, dtype=torch.float64) * 0.1)
            q, _ = torch.linalg.qr(torch.randn((d, d), generator=g, dtype=torch.float64))
            sd[name + ".p"] = q * (0.5 + torch.rand((d,), generator=g, dtype=torch.float64))[None, :]
    d = dim
    for j, h in enumerate(hdims):
        bound = (6.0 / (d + h)) ** 0.5
        sd[f"projector.{3 * j}.weight"] = (torch.rand((h, d), generator=g) * 2 - 1) * bound
        sd[f"projector.{3 * j}.bias"] = torch.randn((h,), generator=g) * 0.02
        if j + 1 < len(hdims):
            sd[f"projector.{3 * j + 1}.weight"] = 1.0 + 0.2 * torch.randn((h,), generator=g)
            sd[f"projector.{3 * j + 1}.bias"] = 0.1 * torch.randn((h,), generator=g)
        d = h
    return sd
