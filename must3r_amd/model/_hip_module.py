"""Shared plumbing of the two HIP-backed modules: context lifetime, weight synchronisation,
operand-dtype policy, stream handling."""
import operator
import os

import torch
import torch.nn as nn

from .. import _lib

# 'fp16w2' = fp16 operands with split weights (W_hi + W_lo): the mode that meets the 1e-3 parity target with the widest margin
# 'fp16wa' = the same with PLAIN fp16 weights in the Mlp Linears (2/3 of the GEMM FLOPs in one MFMA pass): inside the target too
_DT = {"bf16": _lib.BF16, "fp16": _lib.F16, "fp16w2": _lib.F16_W2, "fp16wa": _lib.F16_WA, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}
_TORCH_DT = {_lib.BF16: torch.bfloat16, _lib.F16: torch.float16, _lib.F16_W2: torch.float16, _lib.F16_WA: torch.float16}
PRECISIONS = ("bf16", "fp16", "fp16w2", "fp16wa")
_VERSION_OF = operator.attrgetter("_version")
_DEBUG_WEIGHTS = bool(int(os.environ.get("M3R_DEBUG_WEIGHTS", "0") or 0))


def operand_dtype(precision):
    if precision not in _DT:
        raise ValueError(f"precision must be one of {PRECISIONS}, got {precision!r}")
    return _DT[precision]


def autocast_dtype():
    """dtype the reference would run a Linear in right now (blocks/__init__.py:5-16 get_current_dtype)."""
    try:
        if torch.is_autocast_enabled("cuda"):
            return torch.get_autocast_dtype("cuda")
    except TypeError:  # older torch: no device argument
        if torch.is_autocast_enabled():
            return torch.get_autocast_gpu_dtype()
    except Exception:
        pass
    return None


class HipModule(nn.Module):
    """nn.Module whose parameters are mirrored into a libmust3r_hip context on its device."""

    _part = 0        # _lib.PART_ENCODER / PART_DECODER
    _prefix = ""     # "encoder." / "decoder."

    def _hip_init(self, cfg, precision):
        self.cfg = cfg.validate()
        self.precision = precision
        # BASELINE.json configs[4] "fp8 MFMA attention path": Q, K, V and the softmax numerators enter the attention MFMAs as OCP
        # e4m3 (include/must3r_hip.h MUST3R_ATTN_FP8); GEMMs, softmax, accumulators are untouched.  Off by default: its pointmap
        # error (~1e-2) is outside the 1e-3 target -- tests/test_model_gpu.py::test_fp8_attention_*, DESIGN.md section 4.
        self.attention_fp8 = False
        operand_dtype(precision)
        self._ctx = None
        self._ctx_dev = None
        self._synced = None

    # -- weights -------------------------------------------------------------------------------
    # When do the parameters have to be mirrored into the native context again?  The full fingerprint ((data_ptr, _version) of all
    # ~600 parameters) costs ~0.1-0.3 ms of host time per call -- more than the launches of a 224x224 decoder call.  In eval mode
    # (every inference caller of the reference) the check is:
    #   * an epoch counter bumped by everything that replaces or moves parameters wholesale -- `load_state_dict` and `_apply`
    #     (= .to() / .cuda() / .half() ...) of this module AND of every submodule (hooks installed on the children, so that
    #     `model.blocks_dec[3].half()` or a child's `load_state_dict` is seen);
    #   * the SUM of `_version` over a cached parameter list (rebuilt when the epoch moves): any in-place edit of any parameter --
    #     an optimizer step, `p.mul_()`, `p.data.copy_()` -- changes it (~40 us);
    #   * the data_ptr of the first, the middle and the last parameter.
    # Not seen: a parameter OBJECT replaced on a child (`child.weight = nn.Parameter(...)`) -- call `refresh_weights()`; the switch
    # M3R_DEBUG_WEIGHTS=1 asserts the full fingerprint on every forward to find such a caller.  Training mode keeps the full one.
    def _full_fingerprint(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _fingerprint(self):
        if self.training:
            return self._full_fingerprint()
        self._install_child_hooks()
        ps = getattr(self, "_param_cache", None)
        if ps is None or ps[1] != self._weights_epoch:
            allp = list(self.parameters())
            ps = self._param_cache = (allp, self._weights_epoch, (allp[0], allp[len(allp) // 2], allp[-1]))
        fp = (self._weights_epoch, sum(map(_VERSION_OF, ps[0])), ps[2][0].data_ptr(), ps[2][1].data_ptr(), ps[2][2].data_ptr())
        if _DEBUG_WEIGHTS:
            full = self._full_fingerprint()
            last = getattr(self, "_debug_full", None)
            if last is not None and last[0] == fp and last[1] != full:
                raise AssertionError("must3r_amd: parameters changed without the cheap fingerprint noticing (M3R_DEBUG_WEIGHTS); "
                                     "call refresh_weights() after replacing parameter objects")
            self._debug_full = (fp, full)
        return fp

    _weights_epoch = 0
    _hooked_children = None

    def _bump(self):
        self._weights_epoch = self._weights_epoch + 1

    def _install_child_hooks(self):
        """Make wholesale changes on SUBmodules bump this module's epoch (children are plain nn.Modules)."""
        mods = list(self.modules())
        if self._hooked_children == (id(self), len(mods)):
            return
        import weakref
        me = weakref.ref(self)

        def bump(*_a, **_k):
            m = me()
            if m is not None:
                m._bump()
        for m in mods:
            owner = getattr(m, "_m3r_hooked", None)
            if m is self or (owner is not None and owner() is self):
                continue
            m.register_load_state_dict_post_hook(bump)
            orig = m._apply

            def _apply(fn, *a, _orig=orig, **k):
                bump()
                return _orig(fn, *a, **k)
            object.__setattr__(m, "_apply", _apply)
            object.__setattr__(m, "_m3r_hooked", me)
        self._hooked_children = (id(self), len(mods))   # (a deepcopy carries the original's hooks: its id differs, it installs its own)

    def _apply(self, fn, *a, **k):
        self._bump()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._bump()
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        self._bump()   # the two modes use different fingerprints
        return super().train(mode)

    def refresh_weights(self):
        """Force a re-upload of all parameters on the next forward."""
        self._bump()
        self._synced = None

    def _device_index(self):
        p = next(self.parameters())
        if not p.is_cuda:
            raise RuntimeError(
                "must3r_amd: the model is on %s; the HIP path needs an MI355X (gfx950) device and has no CPU "
                "fallback -- move the module with .to('cuda')" % p.device)
        return p.device.index if p.device.index is not None else torch.cuda.current_device()

    def _context(self):
        dev = self._device_index()
        if self._ctx is None or self._ctx_dev != dev:
            if self._ctx is not None:
                self._ctx.close()
            self._ctx = _lib.Context(self.cfg, dev)
            self._ctx_dev = dev
            self._synced = None
        fp = self._fingerprint()
        if self._synced != fp:
            torch.cuda.synchronize(dev)
            for k, v in self.state_dict().items():
                self._ctx.load_weight(self._prefix + k, v)
            self._ctx.finalize(self._part)
            self._synced = fp
        return self._ctx

    @staticmethod
    def _stream(dev):
        return torch.cuda.current_stream(dev).cuda_stream

    @staticmethod
    def _check_input(t, name, dtype=None):
        if not t.is_cuda:
            raise RuntimeError(f"must3r_amd: `{name}` is on {t.device}; the HIP path has no CPU fallback")
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.contiguous()
