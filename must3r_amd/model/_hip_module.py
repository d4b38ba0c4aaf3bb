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
_DATA_PTR_OF = operator.methodcaller("data_ptr")
_DTYPE_OF = operator.attrgetter("dtype")
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
        # BASELINE.json configs[4] "fp8 MFMA attention path": Q and K enter the Q K^T MFMAs as OCP e4m3 (include/must3r_hip.h MUST3R_ATTN_FP8).
        # PARKED in r06 (+0.4 % at 1.2e-3 ... 1.4e-3 from the 16-bit path): only a library built with `make EXTRA=-DM3R_ATTN_FP8`
        # accepts True here (the property below refuses it otherwise) -- tests/test_model_gpu.py::test_fp8_attention_* then run, else skip.
        self._attention_fp8 = False
        operand_dtype(precision)
        self._ctx = None
        self._ctx_dev = None
        self._synced = None

    @property
    def attention_fp8(self):
        return self._attention_fp8

    @attention_fp8.setter
    def attention_fp8(self, on):
        if on and not _lib.has_fp8_attention():
            raise RuntimeError("attention_fp8: the e4m3 attention path is parked (outside the 1e-3 target for +0.4 %); rebuild libmust3r_hip with "
                               "`make -C must3r_amd/csrc EXTRA=-DM3R_ATTN_FP8` to experiment with it")
        self._attention_fp8 = bool(on)

    # -- weights -------------------------------------------------------------------------------
    # When do the parameters have to be mirrored into the native context again?  The full fingerprint (a tuple of (data_ptr, _version)
    # of all ~600 parameters, rebuilt from `self.parameters()`) costs 0.1-0.3 ms of host time per call.  In eval mode (every inference
    # caller of the reference) the check works on a cached parameter list and three integers instead:
    #   * an epoch counter bumped by what replaces or moves the parameters of THIS module wholesale (`load_state_dict`, `_apply` =
    #     .to() / .cuda() / .half(), `train`); the cached list is rebuilt when it moves;
    #   * the ordered hash of `_version` over the list (r05; a sum through r04): any in-place edit of a parameter tensor -- an optimizer step, `p.mul_()`,
    #     `p.copy_()` under no_grad, a submodule's `load_state_dict` -- changes it;
    #   * the ordered hash of `data_ptr()` (and of the dtypes) over the list: a conversion or move of ANY submodule (`model.blocks_dec[3].half()`,
    #     `child.to(...)`) gives its parameters new storage.  (r03 bumped the epoch from `_apply` wrappers installed on the children's
    #     instances instead; those closures broke `copy.deepcopy(model).half()` and `torch.save(model)` -- ADVICE r03 -- and are gone:
    #     nothing is patched onto any module.)
    # NOT seen -- call `refresh_weights()` after these:
    #   * a CHILD converted and converted back (`model.blocks_dec[i].half().float()`) when the caching allocator hands the same blocks
    #     back: same pointers, same dtypes, same version counters, fp16-rounded values (`param.data = fn(param.data)` keeps `_version`).
    #     Converting the WHOLE module goes through its own `_apply` and is seen;
    #   * edits through `.data` (`p.data.copy_(w)`, `p.data.mul_(..)`, EMA-style updates): `.data` is a detached alias with its own
    #     version counter, so neither sum moves;
    #   * a parameter OBJECT replaced on a child (`child.weight = nn.Parameter(...)`): the cached list still holds the old one.
    # M3R_DEBUG_WEIGHTS=1 asserts the full fingerprint on every forward to find such a caller.  Training mode keeps the full one.
    def _full_fingerprint(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _fingerprint(self):
        if self.training:
            return self._full_fingerprint()
        ps = self.__dict__.get("_param_cache")
        if ps is None or ps[1] != self._weights_epoch:
            ps = (list(self.parameters()), self._weights_epoch)
            self.__dict__["_param_cache"] = ps
        # (r05, ADVICE r04: ORDERED hashes, not sums -- a swap or permutation of storages between parameters keeps a sum -- and the dtypes)
        fp = (self._weights_epoch, hash(tuple(map(_VERSION_OF, ps[0]))), hash(tuple(map(_DATA_PTR_OF, ps[0]))), hash(tuple(map(_DTYPE_OF, ps[0]))))
        if _DEBUG_WEIGHTS:
            full = self._full_fingerprint()
            last = self.__dict__.get("_debug_full")
            if last is not None and last[0] == fp and last[1] != full:
                raise AssertionError("must3r_amd: parameters changed without the cheap fingerprint noticing (M3R_DEBUG_WEIGHTS); "
                                     "call refresh_weights() after replacing parameter objects or editing them through .data")
            self.__dict__["_debug_full"] = (fp, full)
        return fp

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

    # The native context, the cached parameter list and the sync stamp belong to THIS object: a copy (copy.deepcopy, pickle / torch.save)
    # starts without them and mirrors its own parameters on its first forward.
    _TRANSIENT = ("_ctx", "_ctx_dev", "_synced", "_param_cache", "_debug_full")

    def __getstate__(self):
        st = self.__dict__.copy()
        for k in self._TRANSIENT:
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self._ctx = None
        self._ctx_dev = None
        self._synced = None

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._TRANSIENT:
                new.__dict__[k] = copy.deepcopy(v, memo)
        new._ctx = None
        new._ctx_dev = None
        new._synced = None
        return new

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
