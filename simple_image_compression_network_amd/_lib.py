"""ctypes loader for libsicn.so (the HIP implementation behind include/sicn.h).

There is no CPU fallback: if the shared library is missing or does not export a symbol of
include/sicn.h this module raises, loudly, at import of the first entry point that needs it."""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

from .config import CLayerDesc

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SICN_LIB", _PKG / "libsicn.so"))

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


class COptions(ctypes.Structure):
    """ctypes image of `sicn_options` (include/sicn.h): kernel-selection knobs, passed by value per call / per net."""
    _fields_ = [(n, ctypes.c_int32) for n in ("struct_bytes", "force_generic", "mfma_shape", "tile_x", "strip_chunks",
                                              "no_phase_layout", "split_n", "wave_tile", "prefetch", "persistent_grid", "split_k", "l7_loader", "l0_form", "gdn_fuse")] + [("reserved", ctypes.c_int32 * 2)]


def make_options(**kw) -> "COptions":
    """Library defaults (sicn_options_init) with the given fields replaced, e.g. make_options(tile_x=16)."""
    o = COptions()
    lib().sicn_options_init(ctypes.byref(o))
    for k, v in kw.items():
        if k not in dict(COptions._fields_) or k in ("struct_bytes", "reserved"):
            raise AttributeError(f"sicn_options has no field {k}")
        setattr(o, k, int(v))
    return o


# every symbol include/sicn.h declares: (restype, argtypes)
_descp = ctypes.POINTER(CLayerDesc)
ABI = {
    "sicn_version": (_i, []),
    "sicn_has_alt_kernels": (_i, []),
    "sicn_strerror": (ctypes.c_char_p, [_i]),
    "sicn_validate_desc": (_i, [_descp]),
    "sicn_weights_from_finn_tiles": (_i, [_descp, _vp, _i, _vp, ctypes.POINTER(_vp)]),
    "sicn_weights_free": (None, [_vp]),
    "sicn_conv2d": (_i, [_descp, _vp, _vp, _vp, _i, _vp]),
    "sicn_deconv522": (_i, [_descp, _vp, _vp, _vp, _i, _vp]),
    "sicn_kernel_for": (ctypes.c_char_p, [_descp]),
    "sicn_options_init": (None, [ctypes.POINTER(COptions)]),
    "sicn_conv2d_opt": (_i, [_descp, _vp, _vp, _vp, _i, ctypes.POINTER(COptions), _vp]),
    "sicn_deconv522_opt": (_i, [_descp, _vp, _vp, _vp, _i, ctypes.POINTER(COptions), _vp]),
    "sicn_net_create_opt": (_i, [_descp, ctypes.POINTER(_vp), _i, ctypes.POINTER(COptions), ctypes.POINTER(_vp)]),
    "sicn_net_create": (_i, [_descp, ctypes.POINTER(_vp), _i, ctypes.POINTER(_vp)]),
    "sicn_net_free": (None, [_vp]),
    "sicn_net_workspace_bytes": (_sz, [_vp, _i]),
    "sicn_net_forward": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "sicn_eight_layers_net": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "sicn_net_profile": (_i, [_vp, _i]),
    "sicn_net_layer_ms": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_i)]),
    "sicn_crop_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sicn_debug_plan": (_i, [_descp, _i, ctypes.POINTER(COptions), _i, ctypes.POINTER(ctypes.c_int32)]),
    "sicn_debug_xcd_item": (ctypes.c_longlong, [ctypes.c_longlong, ctypes.c_longlong, _i]),
}


class CodecInfo(ctypes.Structure):
    """ctypes image of `sicn_codec_info` (include/sicn_codec.h)."""
    _fields_ = [(n, ctypes.c_uint32) for n in ("mode", "image_width", "image_height", "lat_w", "lat_h", "lat_c",
                                               "n_symbols", "n_streams", "payload_bytes", "adler32", "stream_symbols")]


class CConvLayerDesc(ctypes.Structure):
    """ctypes image of `sicn_convlayer_desc` (include/sicn_convlayer.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in ("K", "IFM_CH", "IFM_DIM", "OFM_CH", "OFM_DIM", "SIMD", "PE", "IN_BIT", "IN_SIGNED",
                                              "W_BIT", "W_TILES", "ACC_BIT", "ACC_SIGNED", "OUT_BIT", "activation", "NUM_TH",
                                              "ACT_VAL")]


_cldp = ctypes.POINTER(CConvLayerDesc)
# include/sicn_convlayer.h (the generic ConvLayer_Batch surface of convlayer.h:89-125)
CONVLAYER_ABI = {
    "sicn_convlayer_validate": (_i, [_cldp]),
    "sicn_convlayer_params_create": (_i, [_cldp, _vp, _i, _vp, ctypes.POINTER(_vp)]),
    "sicn_convlayer_params_free": (None, [_vp]),
    "sicn_conv_layer_batch": (_i, [_cldp, _vp, _vp, _vp, _i, _vp]),
    "sicn_conv_layer_batch_kernel": (_i, [_cldp, _vp, _vp, _vp, _i, _i, _vp]),
}

# include/sicn_gdn.h (extension beyond the reference: fixed-point GDN / IGDN in place of the ReLU)
GDN_ABI = {
    "sicn_gdn_create": (_i, [_i, _i, _i, _vp, _vp, ctypes.POINTER(_vp)]),
    "sicn_gdn_free": (None, [_vp]),
    "sicn_gdn_apply": (_i, [_vp, _vp, ctypes.c_longlong, _vp]),
    "sicn_conv2d_gdn": (_i, [_descp, _vp, _vp, _vp, _vp, _i, ctypes.POINTER(COptions), _vp]),
    "sicn_deconv522_gdn": (_i, [_descp, _vp, _vp, _vp, _vp, _i, ctypes.POINTER(COptions), _vp]),
    "sicn_net_create_gdn": (_i, [_descp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i, ctypes.POINTER(COptions), ctypes.POINTER(_vp)]),
    "sicn_gdn_selftest_roots": (ctypes.c_longlong, [_i, ctypes.c_uint32, ctypes.c_ulonglong]),
    "sicn_gdn_spec_version": (_i, []),
}

_u32 = ctypes.c_uint32
# include/sicn_codec.h (extension beyond the reference: latent container + rANS coder)
CODEC_ABI = {
    "sicn_codec_max_bytes": (_sz, [_i, _u32]),
    "sicn_codec_workspace_bytes": (_sz, [_i, _u32]),
    "sicn_codec_encode": (_i, [_i, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _sz, ctypes.POINTER(_sz), _vp, _sz, _vp]),
    "sicn_codec_parse_header": (_i, [_vp, _sz, ctypes.POINTER(CodecInfo)]),
    "sicn_codec_decode": (_i, [_vp, _sz, _vp, _sz, ctypes.POINTER(CodecInfo), _vp, _sz, _vp]),
    "sicn_codec_batch_workspace_bytes": (_sz, [_i, _u32, _u32]),
    "sicn_codec_encode_batch": (_i, [_i, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _sz, ctypes.POINTER(_sz), _vp, _sz, _vp]),
    "sicn_codec_decode_batch": (_i, [_vp, _sz, ctypes.POINTER(_sz), _u32, _vp, _sz, ctypes.POINTER(CodecInfo), _vp, _sz, _vp]),
    "sicn_codec_encode_batch_async": (_i, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _sz, _vp, _vp, _sz, _vp]),
    "sicn_codec_decode_batch_async": (_i, [_vp, _sz, _vp, _u32, _u32, _u32, _u32, _vp, _sz, _vp, _vp, _sz, _vp]),
    "sicn_codec_max_bytes_sl": (_sz, [_u32, _u32]),
    "sicn_codec_workspace_bytes_sl": (_sz, [_u32, _u32]),
    "sicn_codec_batch_workspace_bytes_sl": (_sz, [_u32, _u32, _u32]),
    "sicn_codec_encode_batch_async_sl": (_i, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _sz, _vp, _vp, _sz, _vp, _u32]),
    "sicn_codec_decode_batch_async_sl": (_i, [_vp, _sz, _vp, _u32, _u32, _u32, _u32, _vp, _sz, _vp, _vp, _sz, _vp, _u32]),
    "sicn_codec_ctx_max_bytes": (_sz, [_u32, _u32, _u32]),
    "sicn_codec_ctx_workspace_bytes": (_sz, [_u32, _u32, _u32, _u32]),
    "sicn_codec_ctx_encode_batch_async": (_i, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _sz, _vp, _vp, _sz, _vp]),
    "sicn_codec_ctx_decode_batch_async": (_i, [_vp, _sz, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _sz, _vp]),
    "sicn_codec_selftest_div": (ctypes.c_longlong, [_u32, _u32, ctypes.POINTER(ctypes.c_ulonglong)]),
}

_lib = None


class SicnError(RuntimeError):
    def __init__(self, code: int, what: str):
        self.code = code
        msg = lib().sicn_strerror(code).decode() if _lib is not None else "?"
        super().__init__(f"{what}: {msg} ({code})")


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {_PKG / 'csrc'}`; there is no non-HIP fallback")
        # PyTorch ships its own copy of the HIP runtime.  Device pointers handed to this library come from
        # torch, so both must live in ONE runtime: load torch's first and let libsicn.so bind to it (loaded
        # the other way round, the first launch fails with a HIP runtime error).
        try:
            import torch  # noqa: F401
        except ImportError:      # symbol checks etc. work without it
            pass
        L = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in {**ABI, **CODEC_ABI, **CONVLAYER_ABI, **GDN_ABI}.items():
            fn = getattr(L, name)          # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise SicnError(code, what)
