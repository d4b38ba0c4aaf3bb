"""ctypes binding of ``libmust3r_hip.so`` (C ABI declared in ``include/must3r_hip.h``).

The shared library is built in-tree by ``must3r_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no CPU or eager-PyTorch fallback: if the library is missing or no gfx950 device is visible
the product path raises, it never silently computes elsewhere.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmust3r_hip.so")

BF16, F16, F16_W2, F16_WA = 0, 1, 2, 3
ATTN_FP8 = 0x100   # OR-able: fp8 (e4m3) attention operands, include/must3r_hip.h
MEM_KV, MEM_NORM_Y, MEM_RAW = 0, 1, 2
PART_ENCODER, PART_DECODER = 1, 2
EPI_STORE16, EPI_STORE16_GELU, EPI_QKV_ROPE, EPI_RESID_F32, EPI_F32, EPI_HEAD = range(6)
ABI_VERSION = 8
ACT_NORM_EXP, ACT_LINEAR = 0, 1

# every symbol include/must3r_hip.h declares
EXPORTS = (
    "must3r_hip_abi_version", "must3r_hip_last_error", "must3r_hip_create", "must3r_hip_destroy",
    "must3r_hip_load_weight", "must3r_hip_finalize_weights", "must3r_hip_encode", "must3r_hip_decode",
    "must3r_hip_postprocess", "must3r_hip_op_gemm", "must3r_hip_rope_table", "must3r_hip_op_attention",
    "must3r_hip_op_layernorm", "must3r_hip_op_im2col", "must3r_hip_op_cast", "must3r_hip_set_profiling",
    "must3r_hip_get_profile", "must3r_hip_debug_tr_probe", "must3r_hip_attention_scratch_bytes",
    "must3r_hip_postprocess_cam", "must3r_hip_postprocess_cam_scratch_bytes",
    "must3r_hip_nn_query", "must3r_hip_quadrant_ids",
    "must3r_hip_affine", "must3r_hip_row_norm", "must3r_hip_l2_normalize", "must3r_hip_layernorm_act_f32", "must3r_hip_topk_gather", "must3r_hip_weighted_spoc",
    "must3r_hip_op_gemm_lnfold",
    "must3r_hip_postprocess_act", "must3r_hip_postprocess_cam_act",
    "must3r_hip_op_sparse24_pack", "must3r_hip_op_gemm_sp",
    "must3r_hip_set_option", "must3r_hip_cp_slot_bytes", "must3r_hip_cp_slot_bytes16", "must3r_hip_op_gemm_fold256", "must3r_hip_has_fp8_attention",
)


class Config(C.Structure):
    _fields_ = [("img_size", C.c_int32), ("patch_size", C.c_int32),
                ("enc_dim", C.c_int32), ("enc_depth", C.c_int32), ("enc_heads", C.c_int32),
                ("dec_dim", C.c_int32), ("dec_depth", C.c_int32), ("dec_heads", C.c_int32),
                ("mlp_ratio", C.c_int32), ("rope_freq", C.c_float), ("rope_f0", C.c_float)]


class Group(C.Structure):
    _fields_ = [("tokens", C.c_void_p), ("pos", C.c_void_p),
                ("n_views", C.c_int32), ("n_tokens", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("pointmaps", C.c_void_p), ("pointmaps_scene_stride", C.c_int64)]


# must3r_hip_cp_exchange_fn: (user, layer, slots, slot_bytes, n_slots, my_slot, stream) -> status
CpExchangeFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)


class Cp(C.Structure):
    """must3r_hip_cp: context-parallel cross attention over a memory sharded across ranks (include/must3r_hip.h, ABI 8)."""
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("n_mem_total", C.c_int32), ("partial16", C.c_int32),
                ("slots", C.c_void_p), ("slot_bytes", C.c_size_t), ("exchange", CpExchangeFn), ("user", C.c_void_p)]


class DecodeArgs(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("mem_mode", C.c_int32), ("render", C.c_int32), ("first_call", C.c_int32),
                ("n_groups", C.c_int32), ("groups", C.POINTER(Group)), ("n_mem", C.c_int32),
                ("mem", C.POINTER(C.c_void_p)), ("feats", C.c_void_p),
                ("mem_capacity", C.c_int32), ("n_scenes", C.c_int32), ("mem_scene_stride", C.c_int64),
                ("cp", C.POINTER(Cp)), ("causal", C.c_int32)]


class ProfRecord(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("ms", C.c_double), ("flops", C.c_double), ("calls", C.c_int64)]


class HipError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen the library (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C must3r_amd/csrc). must3r_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64p, fp = C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_float
    lib.must3r_hip_abi_version.restype = i32
    lib.must3r_hip_last_error.restype = C.c_char_p
    lib.must3r_hip_create.argtypes = [C.POINTER(Config), i32, C.POINTER(vp)]
    lib.must3r_hip_destroy.argtypes = [vp]
    lib.must3r_hip_destroy.restype = None
    lib.must3r_hip_load_weight.argtypes = [vp, C.c_char_p, vp, i32, i32, i64p]
    lib.must3r_hip_finalize_weights.argtypes = [vp, i32]
    lib.must3r_hip_encode.argtypes = [vp, i32, vp, i32, i32, i32, vp, vp, vp]
    lib.must3r_hip_decode.argtypes = [vp, C.POINTER(DecodeArgs), vp]
    lib.must3r_hip_postprocess.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    lib.must3r_hip_postprocess_act.argtypes = [vp, i32, vp, vp, vp, C.c_size_t, vp]
    lib.must3r_hip_op_gemm.argtypes = [i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32,
                                       vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.must3r_hip_rope_table.argtypes = [fp, fp, i32, vp]
    lib.must3r_hip_op_attention.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, vp, i32, vp]
    lib.must3r_hip_attention_scratch_bytes.argtypes = [i32, i32, i32]
    lib.must3r_hip_attention_scratch_bytes.restype = C.c_size_t
    lib.must3r_hip_op_layernorm.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, fp, vp]
    lib.must3r_hip_op_gemm_lnfold.argtypes = [i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, fp, vp, i32, vp, vp, i32, i32, fp, i32, vp]
    lib.must3r_hip_op_gemm_fold256.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, fp, vp, vp, vp, i32, i32, fp, i32, vp]
    lib.must3r_hip_op_sparse24_pack.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.must3r_hip_op_gemm_sp.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp]
    lib.must3r_hip_op_im2col.argtypes = [i32, vp, vp, i32, i32, i32, vp]
    lib.must3r_hip_op_cast.argtypes = [i32, vp, vp, vp, C.c_size_t, vp]
    lib.must3r_hip_debug_tr_probe.argtypes = [vp, vp]
    lib.must3r_hip_set_profiling.argtypes = [vp, i32]
    lib.must3r_hip_postprocess_cam.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.must3r_hip_postprocess_cam_act.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.must3r_hip_affine.argtypes = [i32, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, vp]
    lib.must3r_hip_row_norm.argtypes = [vp, i32, i32, vp, vp]
    lib.must3r_hip_l2_normalize.argtypes = [vp, C.c_int64, i32, C.c_int64, vp, vp]
    lib.must3r_hip_layernorm_act_f32.argtypes = [vp, vp, vp, fp, i32, i32, i32, vp, vp]
    lib.must3r_hip_topk_gather.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.must3r_hip_weighted_spoc.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.must3r_hip_nn_query.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, vp]
    lib.must3r_hip_quadrant_ids.argtypes = [vp, C.c_int64, C.POINTER(C.c_float), i32, vp, vp]
    lib.must3r_hip_postprocess_cam_scratch_bytes.argtypes = [i32, i32, i32]
    lib.must3r_hip_postprocess_cam_scratch_bytes.restype = C.c_size_t
    lib.must3r_hip_get_profile.argtypes = [vp, C.POINTER(ProfRecord), i32, i32]
    lib.must3r_hip_set_option.argtypes = [C.c_char_p, C.c_longlong]
    lib.must3r_hip_cp_slot_bytes.argtypes = [vp, i32]
    lib.must3r_hip_cp_slot_bytes.restype = C.c_size_t
    lib.must3r_hip_cp_slot_bytes16.argtypes = [vp, i32]
    lib.must3r_hip_cp_slot_bytes16.restype = C.c_size_t
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ("must3r_hip_abi_version", "must3r_hip_attention_scratch_bytes",
                                                    "must3r_hip_postprocess_cam_scratch_bytes", "must3r_hip_cp_slot_bytes"):
            fn.restype = i32
    if lib.must3r_hip_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.must3r_hip_abi_version()} != {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise HipError(load().must3r_hip_last_error().decode("utf-8", "replace"))


def has_fp8_attention():
    """Was the library built with the parked e4m3 attention path (make EXTRA=-DM3R_ATTN_FP8; include/must3r_hip.h MUST3R_ATTN_FP8)?"""
    return bool(load().must3r_hip_has_fp8_attention())


def set_option(name, value):
    """Process-wide A/B switch of the library (include/must3r_hip.h ``must3r_hip_set_option``): raises on an unknown
    name or a value outside the switch's range."""
    check(load().must3r_hip_set_option(name.encode(), int(value)))


def make_config(cfg):
    return Config(cfg.img_size, cfg.patch_size, cfg.enc_dim, cfg.enc_depth, cfg.enc_heads,
                  cfg.dec_dim, cfg.dec_depth, cfg.dec_heads, cfg.mlp_ratio, cfg.rope_freq, cfg.rope_f0)


class Context:
    """Owner of one ``must3r_hip_ctx`` (one device, one forward in flight)."""

    def __init__(self, cfg, device_index):
        self.lib = load()
        self.handle = C.c_void_p()
        c = make_config(cfg)
        check(self.lib.must3r_hip_create(C.byref(c), int(device_index), C.byref(self.handle)))
        self.device_index = int(device_index)

    def load_weight(self, name, tensor):
        """tensor: contiguous fp32 torch tensor (host or the context's device)."""
        import torch
        t = tensor.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.float().contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        check(self.lib.must3r_hip_load_weight(self.handle, name.encode(), C.c_void_p(t.data_ptr()),
                                              1 if t.is_cuda else 0, t.dim(), shape))

    def finalize(self, parts):
        check(self.lib.must3r_hip_finalize_weights(self.handle, parts))

    def set_profiling(self, on):
        check(self.lib.must3r_hip_set_profiling(self.handle, 1 if on else 0))

    def get_profile(self, reset=True):
        recs = (ProfRecord * 160)()   # 7 class rows + one row per kernel symbol (names prefixed "k:")
        n = self.lib.must3r_hip_get_profile(self.handle, recs, 160, 1 if reset else 0)
        return {recs[i].name.decode(): {"ms": recs[i].ms, "flops": recs[i].flops, "calls": recs[i].calls}
                for i in range(n)}

    def close(self):
        if self.handle:
            self.lib.must3r_hip_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
