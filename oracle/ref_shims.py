"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (must3r_amd/).

Leaf shims that let the reference's own ``must3r/model/*.py`` be imported *verbatim* from
``/root/reference`` inside the build container.  The reference depends on un-vendored packages
(SURVEY.md section 8c): ``dust3r`` (git submodule, empty dir -- .gitmodules:1-4),
``croco`` (setup.py:36, branch ``croco_module``, no commit pin), ``curope`` (setup.py:4),
``torchvision`` (tools/image.py:6, import only).  None of them can be installed here (no network),
so the leaves are restated below from their call sites in the reference and the upstream
specification recorded in SURVEY.md Appendix A.

    PARITY UNPINNED for these leaves: the reference ships no test, golden vector or pinned commit
    for croco.models.blocks.Mlp / PositionGetter, croco.models.pos_embed.RoPE2D or
    dust3r.patch_embed.*.  Everything *above* the leaves (must3r/model/*.py) is the reference's own
    code, executed unmodified.

This module is used by ``oracle/make_golden.py`` (fixture generation) and by the CPU tests that
cross-check ``oracle/must3r_ref.py`` against the real reference when ``/root/reference`` exists.
It must not be used on the GPU box (the reference tree is not there).
"""
import sys
import types
import math

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


# ----------------------------------------------------------------------------------------------
# croco.models.blocks  (call sites: must3r/model/blocks/layers.py:7, feedback_mechanism.py:8,
#                        decoder.py:561)
# ----------------------------------------------------------------------------------------------
class Mlp(nn.Module):
    """fc2(act(fc1(x))); state-dict keys fc1.*, fc2.* (feedback_mechanism.py:30-31 touches fc2)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class DropPath(nn.Module):
    """Stochastic depth. Never instantiated at inference (layers.py:49,79 build nn.Identity)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


class PositionGetter:
    """(b, h, w, device) -> int64[b, h*w, 2]; row-major tokens, column 0 = y, column 1 = x."""

    def __init__(self):
        self.cache = {}

    def __call__(self, b, h, w, device):
        if (h, w) not in self.cache:
            ys = torch.arange(h, device=device)
            xs = torch.arange(w, device=device)
            self.cache[h, w] = torch.cartesian_prod(ys, xs)
        return self.cache[h, w].view(1, h * w, 2).expand(b, -1, 2).clone()


# ----------------------------------------------------------------------------------------------
# croco.models.pos_embed.RoPE2D  (built at must3r/model/blocks/pos_embed.py:21, applied at
#                                  must3r/model/blocks/attention.py:42-44)
# ----------------------------------------------------------------------------------------------
class RoPE2D(nn.Module):
    """2-D rotary embedding on tokens[B,H,N,D], positions[B,N,2].

    D is split in two halves; the first is rotated by positions[...,0] (y), the second by
    positions[...,1] (x).  Inside a half of size Dh: Dh/2 frequencies w_i = F0 * freq^(-2i/Dh),
    pairs are (i, i+Dh/2) ("rotate-half"): out = t*cos + rotate_half(t)*sin.
    """

    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base = freq
        self.F0 = F0

    def _cos_sin(self, Dh, npos, device, dtype):
        inv_freq = self.F0 / (self.base ** (torch.arange(0, Dh, 2, device=device).float() / Dh))
        t = torch.arange(npos, device=device, dtype=torch.float32)
        ang = torch.outer(t, inv_freq)
        ang = torch.cat((ang, ang), dim=-1)
        return ang.cos().to(dtype), ang.sin().to(dtype)

    @staticmethod
    def _rot_half(x):
        h = x.shape[-1] // 2
        return torch.cat((-x[..., h:], x[..., :h]), dim=-1)

    def _rope1d(self, tok, p, cos, sin):
        c = cos[p][:, None, :, :]
        s = sin[p][:, None, :, :]
        return tok * c + self._rot_half(tok) * s

    def forward(self, tokens, positions):
        assert tokens.shape[-1] % 4 == 0 and positions.ndim == 3 and positions.shape[-1] == 2
        Dh = tokens.shape[-1] // 2
        cos, sin = self._cos_sin(Dh, int(positions.max()) + 1, tokens.device, tokens.dtype)
        ty, tx = tokens.chunk(2, dim=-1)
        ty = self._rope1d(ty, positions[:, :, 0], cos, sin)
        tx = self._rope1d(tx, positions[:, :, 1], cos, sin)
        return torch.cat((ty, tx), dim=-1)


# ----------------------------------------------------------------------------------------------
# dust3r.patch_embed  (call site: must3r/model/encoder.py:10,43,48)
# ----------------------------------------------------------------------------------------------
class PatchEmbedDust3R(nn.Module):
    """Conv2d(3, dim, k=s=patch) -> flatten -> [B,N,dim]; positions of the actual H/p x W/p grid."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        self.img_size = tuple(img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (self.img_size[0] // patch_size, self.img_size[1] // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()
        self.position_getter = PositionGetter()

    def forward(self, x, **kw):
        B, C, H, W = x.shape
        assert H % self.patch_size[0] == 0 and W % self.patch_size[1] == 0
        x = self.proj(x)
        pos = self.position_getter(B, x.size(2), x.size(3), x.device)
        x = x.flatten(2).transpose(1, 2)
        return self.norm(x), pos


class ManyAR_PatchEmbed(PatchEmbedDust3R):
    """Batch stored landscape (W>=H); portrait samples (true_shape h>w) are transposed before proj."""

    def forward(self, img, true_shape):
        B, C, H, W = img.shape
        assert W >= H and true_shape.shape == (B, 2)
        height, width = true_shape.T
        is_landscape = width >= height
        is_portrait = ~is_landscape
        n_tokens = (H // self.patch_size[0]) * (W // self.patch_size[1])
        x = img.new_zeros((B, n_tokens, self.proj.out_channels))
        pos = torch.zeros((B, n_tokens, 2), dtype=torch.int64, device=img.device)
        if is_landscape.any():
            xl = self.proj(img[is_landscape])
            pos[is_landscape] = self.position_getter(1, xl.size(2), xl.size(3), img.device)
            x[is_landscape] = xl.flatten(2).transpose(1, 2)
        if is_portrait.any():
            xp = self.proj(img[is_portrait].swapaxes(-1, -2))
            pos[is_portrait] = self.position_getter(1, xp.size(2), xp.size(3), img.device)
            x[is_portrait] = xp.flatten(2).transpose(1, 2)
        return self.norm(x), pos


def get_patch_embed(name, img_size, patch_size, enc_embed_dim):
    assert name in ("PatchEmbedDust3R", "ManyAR_PatchEmbed"), name
    return {"PatchEmbedDust3R": PatchEmbedDust3R, "ManyAR_PatchEmbed": ManyAR_PatchEmbed}[name](
        img_size, patch_size, 3, enc_embed_dim)


# ----------------------------------------------------------------------------------------------
# module registration
# ----------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave as a package
    sys.modules[name] = m
    return m


def install(reference_root=REFERENCE_ROOT):
    """Register the stub modules and put the reference on sys.path. Idempotent."""
    sys.dont_write_bytecode = True  # do not drop __pycache__ into the read-only reference tree
    if "must3r_ref_shims_installed" in sys.modules:
        return
    _mod("dust3r")
    _mod("dust3r.utils")
    _mod("dust3r.utils.path_to_croco")
    _mod("dust3r.utils.geometry", geotrf=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()))
    _mod("dust3r.utils.misc",
         invalid_to_zeros=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()),
         invalid_to_nans=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()))
    # retrieval/model.py:12 imports the image loader at module level; only its dataset class calls it (image files)
    _mod("dust3r.utils.image", load_images=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()))
    _mod("dust3r.patch_embed", get_patch_embed=get_patch_embed, PatchEmbedDust3R=PatchEmbedDust3R,
         ManyAR_PatchEmbed=ManyAR_PatchEmbed)
    try:
        from . import cam_ref
    except ImportError:  # imported as a top-level module (oracle/ on sys.path)
        import cam_ref
    # restated third-party leaves of postprocess(compute_cam=True) (engine/inference.py:13,7): see oracle/cam_ref.py
    _mod("dust3r.post_process", estimate_focal_knowing_depth=cam_ref.estimate_focal_knowing_depth)
    _mod("croco")
    _mod("croco.models")
    _mod("croco.models.blocks", Mlp=Mlp, DropPath=DropPath, PositionGetter=PositionGetter)
    _mod("croco.models.pos_embed", RoPE2D=RoPE2D)
    try:
        import torchvision  # noqa: F401
    except Exception:
        tv = _mod("torchvision")
        tv.transforms = _mod("torchvision.transforms")
    try:
        import roma  # noqa: F401
    except Exception:
        _mod("roma", rigid_points_registration=cam_ref.rigid_points_registration,
             special_procrustes=cam_ref.special_procrustes)
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    sys.modules["must3r_ref_shims_installed"] = types.ModuleType("must3r_ref_shims_installed")


def import_reference_model(reference_root=REFERENCE_ROOT):
    """Return the reference's own ``must3r.model`` module (verbatim code from /root/reference)."""
    install(reference_root)
    import must3r.model as ref_model  # noqa: E402
    return ref_model


def build_reference(cfg, sd_enc, sd_dec, memory_mode="kv", feedback_type="single_mlp"):
    """Reference constructors (encoder.py:13, decoder.py:14) for ``cfg`` (must3r_amd.config.ModelConfig),
    loaded ``strict=True`` with the given state dicts -- which also pins the state-dict key contract
    of SURVEY.md section 8b."""
    ref = import_reference_model()
    enc = ref.Dust3rEncoder(img_size=(cfg.img_size, cfg.img_size), patch_size=cfg.patch_size,
                            embed_dim=cfg.enc_dim, depth=cfg.enc_depth, num_heads=cfg.enc_heads)
    dec = ref.MUSt3R(img_size=(cfg.img_size, cfg.img_size), enc_embed_dim=cfg.enc_dim,
                     patch_size=cfg.patch_size, embed_dim=cfg.dec_dim, output_dim=cfg.output_dim,
                     depth=cfg.dec_depth, num_heads=cfg.dec_heads, feedback_type=feedback_type,
                     memory_mode=memory_mode, landscape_only=False)
    enc.load_state_dict(sd_enc, strict=True)
    dec.load_state_dict(sd_dec, strict=True)
    return enc.eval(), dec.eval()


def build_reference_causal(cfg, sd_dec, memory_mode="kv", feedback_type="single_mlp", **causal_kw):
    """The reference's CausalMUSt3R (decoder.py:352-553: the class the released checkpoints were TRAINED as -- one forward over a sequence of views in which view i
    cross-attends the memory of the views before it) with the same state dict (it adds no parameters); ``causal_kw``: protected_imgs, use_mem_mask, ..."""
    import importlib
    import_reference_model()
    dmod = importlib.import_module("must3r.model.decoder")
    dec = dmod.CausalMUSt3R(img_size=(cfg.img_size, cfg.img_size), enc_embed_dim=cfg.enc_dim,
                            patch_size=cfg.patch_size, embed_dim=cfg.dec_dim, output_dim=cfg.output_dim,
                            depth=cfg.dec_depth, num_heads=cfg.dec_heads, feedback_type=feedback_type,
                            memory_mode=memory_mode, landscape_only=False, **causal_kw)
    dec.load_state_dict(sd_dec, strict=True)
    return dec.eval()
