"""Shared plumbing of the two HIP-backed modules: context lifetime, weight synchronisation,
operand-dtype policy, stream handling."""
import torch
import torch.nn as nn

from .. import _lib

# 'fp16w2' = fp16 operands with split weights (W_hi + W_lo): the mode that meets the 1e-3 parity target with margin
_DT = {"bf16": _lib.BF16, "fp16": _lib.F16, "fp16w2": _lib.F16_W2, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}
_TORCH_DT = {_lib.BF16: torch.bfloat16, _lib.F16: torch.float16, _lib.F16_W2: torch.float16}
PRECISIONS = ("bf16", "fp16", "fp16w2")


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
    # When do the parameters have to be mirrored into the native context again?  Walking all ~600 parameters per forward
    # ((data_ptr, _version) each) costs ~0.3 ms of host time per call -- more than the launches of a 224x224 decoder call.
    # In eval mode (every inference caller of the reference) the check is O(1): an epoch counter bumped by everything that
    # replaces or moves parameters wholesale (`load_state_dict`, `_apply` = .to() / .cuda() / .half() ...) plus the
    # (data_ptr, _version) of the first and the last parameter as sentinels (an optimizer step or a bulk in-place edit changes
    # them).  In training mode the full fingerprint is kept.  A caller that edits single parameters in place in eval mode
    # calls `refresh_weights()`.
    def _fingerprint(self):
        if self.training:
            return tuple((p.data_ptr(), p._version) for p in self.parameters())
        ps = getattr(self, "_param_ends", None)
        if ps is None or ps[2] != self._weights_epoch:
            allp = list(self.parameters())
            ps = self._param_ends = (allp[0], allp[-1], self._weights_epoch)
        return (self._weights_epoch, ps[0].data_ptr(), ps[0]._version, ps[1].data_ptr(), ps[1]._version)

    _weights_epoch = 0

    def _bump(self):
        self._weights_epoch = self._weights_epoch + 1

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
