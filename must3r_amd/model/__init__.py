"""``must3r_amd.model`` -- mirrors the public surface of ``must3r.model`` (must3r/model/__init__.py) for the
multi-view forward path: ``Dust3rEncoder``, ``MUSt3R``, ``load_model``, ``get_dtype``,
``get_pointmaps_activation``, ``convert_decoder_args``, ``set_image_size_in_args``, ``ActivationType``,
``apply_activation``.
"""
import re

import torch

from .blocks import ActivationType, MEMORY_MODES  # noqa: F401
from .encoder import Dust3rEncoder  # noqa: F401
from .decoder import MUSt3R, CausalMUSt3R  # noqa: F401


def apply_activation(xyz, activation):
    """blocks/head.py:13-21 + tools/geometry.py:14-18 (torch ops; the fused device kernel is
    ``must3r_amd.engine.postprocess``)."""
    if isinstance(activation, str):
        activation = ActivationType(activation)
    if activation == ActivationType.NORM_EXP:
        d = xyz.norm(dim=-1, keepdim=True)
        return xyz / d.clip(min=1e-8) * torch.expm1(d)
    if activation == ActivationType.LINEAR:
        return xyz
    raise ValueError(f"Unknown activation: {activation}")


def get_pointmaps_activation(decoder, verbose=True):  # model/__init__.py:8-15
    try:
        act = decoder.pointmaps_activation
    except Exception:
        act = ActivationType.NORM_EXP
    if verbose:
        print(f"pointmaps_activation set to {act}")
    return act


def get_dtype(amp):  # model/__init__.py:18-27
    if amp == "fp16":
        return torch.float16
    if amp == "bf16":
        return torch.bfloat16
    assert not amp
    return torch.float32


def convert_decoder_args(decoder_args):  # model/__init__.py:53-63
    decoder_args = decoder_args.replace(" ", "")
    decoder_args = decoder_args.replace("CausalMUSt3R", "MUSt3R").replace("landscape_only=True", "landscape_only=False")
    if "landscape_only=False" not in decoder_args:
        decoder_args = decoder_args[:-1] + ",landscape_only=False)"
    return decoder_args


def set_image_size_in_args(model_args, img_size, verbose=True):  # model/__init__.py:66-108
    """Rewrite ``img_size=(h,h)`` and the RoPE spec so a checkpoint runs at another size with rescaled
    frequencies ('RoPE100' -> 'RoPE100_<trained>:<new>')."""
    model_args = model_args.replace(" ", "")
    m = re.search(r"img_size=\((\d+),(\d+)\)", model_args)
    if not m:
        raise ValueError("No image_size tuple found in model args")
    h, w = (int(v) for v in m.groups())
    assert h == w
    pos_is_arg = True
    ma = re.search(r"pos_embed='([A-Za-z]+)(\d+)\_(\d+):(\d+)'", model_args)
    if ma:
        prefix, freq, base_size, new_size = ma.group(1), int(ma.group(2)), int(ma.group(3)), int(ma.group(4))
    else:
        mb = re.search(r"pos_embed='([A-Za-z]+)(\d+)'", model_args)
        if mb:
            prefix, freq = mb.group(1), int(mb.group(2))
        else:
            prefix, freq, pos_is_arg = "RoPE", 100, False
        base_size = new_size = h
    if verbose:
        print(f"image_size {h} -> {img_size}; pos_embed {prefix}{freq}, base size = {base_size}")
    if img_size != h:
        model_args = model_args.replace(f"img_size=({h},{h})", f"img_size=({img_size},{img_size})")
    if img_size != new_size:
        spec = f"{prefix}{freq}_{base_size}:{img_size}"
        if pos_is_arg:
            model_args = re.sub(r"(pos_embed=')(?:[A-Za-z]+\d+(?:_\d+:\d+)?)(')", rf"\g<1>{spec}\g<2>", model_args)
        else:
            model_args = model_args[:-1] + ",pos_embed='" + spec + "')"
    return model_args


def load_model(chkpt_path, encoder=None, decoder=None, device="cuda", img_size=None, memory_mode=None, verbose=True):
    """model/__init__.py:30-50: checkpoint -> (encoder, decoder) on ``device``, eval mode.  The checkpoint's
    ``args.encoder`` / ``args.decoder`` are Python constructor strings, evaluated here against the HIP-backed
    classes of this package (same names as the reference's)."""
    ckpt = torch.load(chkpt_path, map_location="cpu", weights_only=False)
    enc_args = encoder or ckpt["args"].encoder
    dec_args = decoder or convert_decoder_args(ckpt["args"].decoder)
    if img_size is not None:
        enc_args = set_image_size_in_args(enc_args, img_size, verbose=verbose)
        dec_args = set_image_size_in_args(dec_args, img_size, verbose=verbose)
    ns = {"Dust3rEncoder": Dust3rEncoder, "MUSt3R": MUSt3R, "CausalMUSt3R": CausalMUSt3R, "ActivationType": ActivationType,
          "torch": torch}
    enc = eval(enc_args, ns)  # noqa: S307 -- same contract as the reference (model/__init__.py:38-39)
    dec = eval(dec_args, ns)  # noqa: S307
    if memory_mode is not None:
        dec.change_memory_mode(memory_mode)
    enc.load_state_dict(ckpt["encoder"], strict=True)
    dec.load_state_dict(ckpt["decoder"], strict=True)
    enc.to(device)
    dec.to(device)
    return enc.eval(), dec.eval()
