"""ctypes binding of libmspi_hip.so (C ABI declared in include/mspi_hip.h).

There is no CPU fallback: if the shared library is missing or a kernel launch fails the
call raises.  `build()` (mspi_amd/build.py) compiles the library in-tree with hipcc.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSPI_LIB_PATH") or os.path.join(_HERE, "csrc", "libmspi_hip.so")   # override: A/B builds of the library

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID, ACT_SWISH = range(5)
PREC_F32, PREC_F16X3 = 0, 1


class MspiError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("T", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
        ("sN", C.c_int64), ("sT", C.c_int64), ("sH", C.c_int64), ("sW", C.c_int64), ("sC", C.c_int64),
        ("kT", C.c_int32), ("kH", C.c_int32), ("kW", C.c_int32),
        ("strT", C.c_int32), ("strH", C.c_int32), ("strW", C.c_int32),
        ("padT", C.c_int32), ("padH", C.c_int32), ("padW", C.c_int32),
        ("To", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("Cout", C.c_int32),
        ("ldy", C.c_int64), ("ldw", C.c_int64), ("ldr", C.c_int64),
        ("act", C.c_int32),
        ("prec", C.c_int32),
        ("w_scale", C.c_float),
        ("tile", C.c_int32),
        ("w_blocked", C.c_void_p),
    ]


class DwConvDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("T", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
        ("ldx", C.c_int64), ("ldy", C.c_int64),
        ("kT", C.c_int32), ("kH", C.c_int32), ("kW", C.c_int32),
        ("strT", C.c_int32), ("strH", C.c_int32), ("strW", C.c_int32),
        ("padT", C.c_int32), ("padH", C.c_int32), ("padW", C.c_int32),
        ("To", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("act", C.c_int32),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("Hh", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32), ("D", C.c_int32),
        ("Dv", C.c_int32), ("nmask", C.c_int32), ("nwin", C.c_int32),
        ("q_sB", C.c_int64), ("q_sH", C.c_int64), ("q_sT", C.c_int64),
        ("k_sB", C.c_int64), ("k_sH", C.c_int64), ("k_sT", C.c_int64),
        ("v_sB", C.c_int64), ("v_sH", C.c_int64), ("v_sT", C.c_int64),
        ("o_sB", C.c_int64), ("o_sH", C.c_int64), ("o_sT", C.c_int64),
        ("scale", C.c_float), ("prec", C.c_int32),
    ]


class MvitAugDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("heads", C.c_int32), ("Dh", C.c_int32), ("DA", C.c_int32),
        ("qT", C.c_int32), ("qH", C.c_int32), ("qW", C.c_int32), ("kT", C.c_int32), ("kH", C.c_int32), ("kW", C.c_int32),
        ("ldq", C.c_int64), ("ldk", C.c_int64),
        ("scale", C.c_float),
    ]


class MlpDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("C", C.c_int32), ("hidden", C.c_int32),
        ("ldx", C.c_int64), ("ldr", C.c_int64), ("ldy", C.c_int64),
        ("ln", C.c_int32), ("act", C.c_int32), ("eps", C.c_float),
        ("w1_scale", C.c_float), ("w2_scale", C.c_float),
    ]


class RowGemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("K", C.c_int32), ("N", C.c_int32),
        ("ldx", C.c_int64), ("ldr", C.c_int64), ("ldy", C.c_int64), ("ldg", C.c_int64),
        ("act", C.c_int32), ("rows_per_sample", C.c_int32), ("w_scale", C.c_float),
    ]


class X3dCaDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("D", C.c_int32), ("Cx", C.c_int32),
        ("ldu", C.c_int64), ("ldr", C.c_int64), ("ldy", C.c_int64), ("ldt", C.c_int64), ("ldg", C.c_int64),
        ("rows_per_sample", C.c_int32), ("wc_scale", C.c_float), ("wa_scale", C.c_float),
    ]


class X3dAbDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("T", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("Cin", C.c_int32), ("Cmid", C.c_int32),
        ("ldx", C.c_int64), ("ldu", C.c_int64),
        ("act", C.c_int32), ("wa_scale", C.c_float),
    ]


class PermuteDesc(C.Structure):
    _fields_ = [("dims", C.c_int32 * 6), ("strides", C.c_int64 * 6), ("src_elems", C.c_int64)]


_P = C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "mspi_version": (C.c_int, []),
    "mspi_last_error": (C.c_char_p, []),
    "mspi_device_count": (C.c_int, []),
    "mspi_set_status_word": (C.c_int, [_P]),
    "mspi_conv_last_config": (C.c_int, []),
    "mspi_conv_splitk_ws_bytes": (C.c_size_t, [C.POINTER(ConvDesc), C.c_int32]),
    "mspi_conv_splitk_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "mspi_conv_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "mspi_dwconv_fwd": (C.c_int, [C.POINTER(DwConvDesc), _P, _P, _P, _P, _P, _P]),
    "mspi_dwconv_pool_rows": (C.c_int, [C.POINTER(DwConvDesc)]),
    "mspi_se_gate": (C.c_int, [_P, C.c_int32, C.c_float, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mspi_layernorm_fwd": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, _P, _P, C.c_float, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mspi_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mspi_attn_ws_bytes": (C.c_size_t, [C.POINTER(AttnDesc)]),
    "mspi_attn_fwd_ws": (C.c_int, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mspi_space_to_depth": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mspi_mvit_qk_augment": (C.c_int, [C.POINTER(MvitAugDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "mspi_mvit_qk_augment_p": (C.c_int, [C.POINTER(MvitAugDesc), _P, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "mspi_maxpool_fwd": (C.c_int, [C.POINTER(DwConvDesc), _P, _P, _P]),
    "mspi_upsample_fwd": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, _P]),
    "mspi_rowgate": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int32, _P]),
    "mspi_logsumexp_sub": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "mspi_mean_rows": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mspi_mean_rows_slices": (C.c_int, [C.c_int32]),
    "mspi_mean_rows_ws": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mspi_neg_cosine": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P]),
    "mspi_add": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "mspi_permute_fwd": (C.c_int, [C.POINTER(PermuteDesc), _P, _P, _P]),
    "mspi_gated_sum_fwd": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P]),
    "mspi_logspec_fwd": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int32, _P, _P, C.c_int32, _P]),
    "mspi_resize_norm_fwd": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int32,
                                        _P, _P, C.c_int32, _P, _P, _P]),
    "mspi_layernorm_sp_fwd": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, _P, _P, C.c_float, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, _P]),
    "mspi_split_planes_fwd": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32, _P, C.c_int64, C.c_int64, _P]),
    "mspi_join_planes_fwd": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _P, C.c_int64, _P]),
    "mspi_gemm_sp_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, _P]),
    "mspi_saliency_metrics": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mspi_rowgemm_packed_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "mspi_rowgemm_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "mspi_rowgemm_fwd": (C.c_int, [C.POINTER(RowGemmDesc), _P, _P, _P, _P, _P, _P, _P]),
    "mspi_x3d_ca_packed_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "mspi_x3d_ca_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "mspi_x3d_ca_fwd": (C.c_int, [C.POINTER(X3dCaDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mspi_x3d_ab_supported": (C.c_int, [C.POINTER(X3dAbDesc)]),
    "mspi_x3d_ab_pool_rows": (C.c_int, [C.POINTER(X3dAbDesc)]),
    "mspi_x3d_ab_packed_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "mspi_x3d_ab_fwd": (C.c_int, [C.POINTER(X3dAbDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "mspi_mlp_packed_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "mspi_mlp_fwd": (C.c_int, [C.POINTER(MlpDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mspi_postprocess_workspace": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mspi_postprocess_u8": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
}

EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load libmspi_hip.so; raises MspiError (never falls back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MspiError(
            "libmspi_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(mspi_amd has no CPU fallback)" % LIB_PATH)
    # One HIP runtime per process: the library is linked against libamdhip64.so.7 and PyTorch ships its own copy under
    # the same SONAME.  Whichever is loaded first serves both; if this library came first, torch would later find the
    # system runtime already resident beside its bundled HSA stack and every launch from here fails with "no
    # ROCm-capable device is detected" (seen when build() loaded the library before anything had imported torch).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.mspi_version() != 2:
        raise MspiError("libmspi_hip.so ABI version %d != 2" % lib.mspi_version())
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mspi_last_error().decode("utf-8", "replace")
        raise MspiError("%s failed (%d): %s" % (what, rc, msg))
