"""ctypes binding of the C ABI declared in include/efm_hip.h.

The HIP library is the ONLY compute path of this package: if `libefm_hip.so` is missing the import of
anything that needs it raises (there is no torch / CPU fallback — the CPU oracle lives in `oracle/` and
is test infrastructure only).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EFM_LIB_PATH") or os.path.join(HERE, "libefm_hip.so")  # EFM_LIB_PATH: A/B runs of two builds (tools/)

EFM_OK = 0
MFM_ORDER_GROUP = 0
MFM_ORDER_RES = 1
L2_ROW = 0
L2_FROBENIUS = 1


class EfmError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    """Mirror of `efm_conv_desc` (include/efm_hip.h)."""

    _fields_ = [(n, c_int32) for n in (
        "batch", "hin", "win", "cin", "cin_p", "hout", "wout", "cout", "cout_p",
        "kh", "kw", "pad_h", "pad_w", "n_pad16", "k_pad", "dn_pad16", "dk_pad", "tune_fwd", "tune_dgrad", "tune_wgrad")]


# name -> (restype, argtypes); the single source the symbol-export test checks against the header.
SIGNATURES = {
    "efm_version": (c_int, []),
    "efm_last_error_string": (c_char_p, []),
    "efm_conv_desc_init": (c_int, [POINTER(ConvDesc)] + [c_int] * 9),
    "efm_conv_weight_elems": (c_size_t, [POINTER(ConvDesc)]),
    "efm_conv_dgrad_weight_elems": (c_size_t, [POINTER(ConvDesc)]),
    "efm_conv_wgrad_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "efm_conv_pack_weights": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p]),
    "efm_conv_unpack_weights": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p]),
    "efm_conv_make_dgrad_weights": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p]),
    "efm_conv_fwd": (c_int, [POINTER(ConvDesc)] + [c_void_p] * 6),
    "efm_conv_mfm_supported": (c_int, [POINTER(ConvDesc)]),
    "efm_conv_mfm_fwd": (c_int, [POINTER(ConvDesc)] + [c_void_p] * 5 + [c_int] * 3 + [c_void_p]),
    "efm_mfm_pool_bwd": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "efm_conv_bwd_data": (c_int, [POINTER(ConvDesc)] + [c_void_p] * 5),
    "efm_conv_bwd_weight": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "efm_conv_bwd_weight_slabs": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "efm_conv_bwd_weight_finish": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "efm_convb_weight_elems": (c_size_t, [POINTER(ConvDesc)]),
    "efm_convb_dgrad_weight_elems": (c_size_t, [POINTER(ConvDesc)]),
    "efm_convb_wgrad_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "efm_nchw_to_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "efm_convb_cast_weights": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p]),
    "efm_convb_fwd": (c_int, [POINTER(ConvDesc)] + [c_void_p] * 6),
    "efm_convb_mfm_fwd": (c_int, [POINTER(ConvDesc)] + [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "efm_convb_bwd_data": (c_int, [POINTER(ConvDesc)] + [c_void_p] * 5),
    "efm_convb_mfm_pool_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 6 + [c_void_p]),
    "efm_convb_bwd_weight": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "efm_convb_mfm_bwd_weight_supported": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "efm_convb_mfm_bwd_weight": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                         c_void_p]),
    "efm_wino_supported": (c_int, [POINTER(ConvDesc)]),
    "efm_wino_u_elems": (c_size_t, [POINTER(ConvDesc), c_int]),
    "efm_wino_make_u": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_int, c_void_p]),
    "efm_wino_make_u_batch": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "efm_wino_fwd": (c_int, [POINTER(ConvDesc)] + [c_void_p] * 6),
    "efm_wino_bwd_data": (c_int, [POINTER(ConvDesc)] + [c_void_p] * 5),
    "efm_wino_wgrad_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "efm_wino_bwd_weight": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "efm_wino_mfm_u_elems": (c_size_t, [POINTER(ConvDesc), c_int]),
    "efm_wino_mfm_make_u": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_int, c_void_p]),
    "efm_wino_mfm_fwd": (c_int, [POINTER(ConvDesc)] + [c_void_p] * 5 + [c_int] * 3 + [c_void_p]),
    "efm_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "efm_rowpack_nchw": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "efm_crop_mirror_u8": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_float, c_void_p]),
    "efm_nhwc_to_nchw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "efm_mfm_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "efm_mfm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "efm_mfmb_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "efm_mfmb_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "efm_maxpool2_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "efm_maxpool2_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "efm_l2norm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "efm_l2norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "efm_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "efm_triplet_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "efm_triplet_bwd": (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_void_p]),
    "efm_triplet_indexed_fwd": (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_float, c_void_p]),
    "efm_triplet_indexed_bwd": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_void_p]),
    "efm_cosine_pairs": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "efm_pair_distance": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "efm_gallery_scores": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "efm_gram_cosine": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "efm_mine_semihard": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "efm_pred_create": (c_int, [c_char_p, c_void_p, c_int, c_int, ctypes.c_uint32, POINTER(c_char_p), POINTER(ctypes.c_uint32),
                               POINTER(ctypes.c_uint32), POINTER(c_void_p)]),
    "efm_pred_set_input": (c_int, [c_void_p, c_char_p, c_void_p, ctypes.c_uint32]),
    "efm_pred_forward": (c_int, [c_void_p]),
    "efm_pred_get_output_shape": (c_int, [c_void_p, ctypes.c_uint32, POINTER(POINTER(ctypes.c_uint32)), POINTER(ctypes.c_uint32)]),
    "efm_pred_get_output": (c_int, [c_void_p, ctypes.c_uint32, c_void_p, ctypes.c_uint32]),
    "efm_pred_free": (c_int, [c_void_p]),
    "efm_conv_kernel_info": (c_int, [POINTER(ConvDesc), c_int, c_int, c_int, ctypes.c_char_p, c_size_t, POINTER(ctypes.c_double)]),
    "efm_sgd_update": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_void_p]),
    "efm_adam_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64] + [c_float] * 6 + [c_int, c_void_p]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises EfmError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EfmError(
            "HIP extension %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # torch first: it ships its own libamdhip64; loading ours before it would bind this library to a second HIP runtime
    # in the same process (seen as "no ROCm-capable device is detected" at the first launch).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != EFM_OK:
        msg = load().efm_last_error_string()
        raise EfmError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def conv_desc(batch, hin, win, cin, cout, kh, kw, pad_h, pad_w):
    d = ConvDesc()
    check(load().efm_conv_desc_init(ctypes.byref(d), batch, hin, win, cin, cout, kh, kw, pad_h, pad_w), "efm_conv_desc_init")
    return d


def pad4(c):
    return (c + 3) & ~3


def pad16(c):
    return (c + 15) & ~15
