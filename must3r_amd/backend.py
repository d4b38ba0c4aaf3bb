"""Back-end switch inside an unmodified naver/must3r checkout -- the ``toggle_memory_efficient_attention`` pattern
(must3r/model/blocks/attention.py:18-27: a module-global flag flipped once at start-up by every entry point), applied to
``must3r.model.load_model`` (model/__init__.py:30-50).

    import must3r_amd.backend as hb
    hb.install()                       # adds toggle_hip_backend / is_hip_backend_enabled to must3r.model, wraps load_model
    must3r.model.toggle_hip_backend(True)
    encoder, decoder = must3r.model.load_model(ckpt, device="cuda")    # HIP-backed modules (must3r_amd.model)

Every caller of the reference (engine/inference.py, slam/model.py:10, demo/*) then runs unmodified on the native modules;
modules that did ``from must3r.model import load_model`` before ``install()`` are re-pointed too.  Nothing of the reference
is copied or edited: the wrapper only dispatches.  ``uninstall()`` restores the original function.
"""
import sys

_STATE = {"enabled": False, "orig": None, "wrapper": None}


def toggle_hip_backend(enabled=True):
    _STATE["enabled"] = bool(enabled)


def is_hip_backend_enabled():
    return _STATE["enabled"]


def _repoint(old, new):
    for name, mod in list(sys.modules.items()):
        if mod is not None and (name == "must3r" or name.startswith("must3r.")) and getattr(mod, "load_model", None) is old:
            setattr(mod, "load_model", new)


def install(enable=None):
    """Wrap ``must3r.model.load_model``.  ``must3r`` must be importable (the reference checkout on ``sys.path``)."""
    import must3r.model as ref_model
    if _STATE["wrapper"] is not None and ref_model.load_model is _STATE["wrapper"]:
        if enable is not None:
            toggle_hip_backend(enable)
        return ref_model
    orig = ref_model.load_model

    def load_model(chkpt_path, encoder=None, decoder=None, device="cuda", img_size=None, memory_mode=None, verbose=True):
        if _STATE["enabled"]:
            from . import model as hip_model
            return hip_model.load_model(chkpt_path, encoder, decoder, device, img_size, memory_mode, verbose)
        return orig(chkpt_path, encoder, decoder, device, img_size, memory_mode, verbose)

    load_model.__doc__ = orig.__doc__
    load_model.__wrapped__ = orig
    _STATE["orig"], _STATE["wrapper"] = orig, load_model
    _repoint(orig, load_model)
    ref_model.load_model = load_model
    ref_model.toggle_hip_backend = toggle_hip_backend
    ref_model.is_hip_backend_enabled = is_hip_backend_enabled
    if enable is not None:
        toggle_hip_backend(enable)
    return ref_model


def uninstall():
    if _STATE["wrapper"] is None:
        return
    import must3r.model as ref_model
    _repoint(_STATE["wrapper"], _STATE["orig"])
    ref_model.load_model = _STATE["orig"]
    for n in ("toggle_hip_backend", "is_hip_backend_enabled"):
        if hasattr(ref_model, n):
            delattr(ref_model, n)
    _STATE.update(enabled=False, orig=None, wrapper=None)
