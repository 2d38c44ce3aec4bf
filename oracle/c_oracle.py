"""ORACLE (test infrastructure) — ctypes binding of oracle/libsicn_oracle.so (sicn_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


class OrDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "K", "S", "P", "IFM_CH", "IFM_ROW", "IFM_COL", "OFM_CH", "OFM_ROW", "OFM_COL",
        "SIMD", "PE", "IN_BIT", "OUT_BIT", "W_BIT", "W_TILES", "transposed")]


def build(force: bool = False) -> Path:
    so = _HERE / "libsicn_oracle.so"
    srcs = [_HERE / n for n in ("sicn_oracle.c", "sicn_codec_oracle.c", "sicn_gdn_oracle.c", "sicn_hyper_oracle.c")]
    if force or not so.exists() or so.stat().st_mtime < max(f.stat().st_mtime for f in srcs):
        subprocess.run(["make", "-C", str(_HERE), "-B", "libsicn_oracle.so"], check=True,
                       capture_output=True)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(str(build()))
        p = ctypes.c_void_p
        for name in ("sicn_or_conv2d_dataflow", "sicn_or_deconv522_dataflow"):
            getattr(L, name).argtypes = [ctypes.POINTER(OrDesc), p, p, p, p, ctypes.c_int]
            getattr(L, name).restype = ctypes.c_int
        for name in ("sicn_or_naive_conv2d", "sicn_or_naive_deconv2d"):
            getattr(L, name).argtypes = [ctypes.POINTER(OrDesc), p, p, p, p]
            getattr(L, name).restype = ctypes.c_int
        L.sicn_or_layer_direct.argtypes = [ctypes.POINTER(OrDesc), p, p, p, p, ctypes.c_int]
        L.sicn_or_layer_direct.restype = ctypes.c_int
        L.sicn_or_layer_direct_act.argtypes = [ctypes.POINTER(OrDesc), p, p, p, p, ctypes.c_int, ctypes.c_int]
        L.sicn_or_layer_direct_act.restype = ctypes.c_int
        L.sicn_or_gdn.argtypes = [p, p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, p, p]
        L.sicn_or_gdn.restype = ctypes.c_int
        L.sicl_or_ctx_max_bytes.argtypes = [ctypes.c_uint32] * 3
        L.sicl_or_ctx_max_bytes.restype = ctypes.c_size_t
        L.sicl_or_ctx_encode.argtypes = [p, p] + [ctypes.c_uint32] * 5 + [p, ctypes.c_size_t]
        L.sicl_or_ctx_encode.restype = ctypes.c_longlong
        L.sicl_or_ctx_decode.argtypes = [p, ctypes.c_size_t, p, p, ctypes.c_size_t, p]
        L.sicl_or_ctx_decode.restype = ctypes.c_longlong
        L.sicn_or_swg_nonsquare_fsm.argtypes = [p, ctypes.c_longlong, p] + [ctypes.c_int] * 10
        L.sicn_or_swg_nonsquare_fsm.restype = ctypes.c_longlong
        L.sicn_or_im2col_s1.argtypes = [p, p] + [ctypes.c_int] * 5
        L.sicn_or_im2col_s1.restype = None
        L.sicl_or_max_bytes.argtypes = [ctypes.c_int, ctypes.c_uint32]
        L.sicl_or_max_bytes.restype = ctypes.c_size_t
        L.sicl_or_encode.argtypes = [ctypes.c_int, p] + [ctypes.c_uint32] * 5 + [p, ctypes.c_size_t]
        L.sicl_or_encode.restype = ctypes.c_longlong
        L.sicl_or_max_bytes_sl.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
        L.sicl_or_max_bytes_sl.restype = ctypes.c_size_t
        L.sicl_or_encode_sl.argtypes = [ctypes.c_int, p] + [ctypes.c_uint32] * 5 + [p, ctypes.c_size_t, ctypes.c_uint32]
        L.sicl_or_encode_sl.restype = ctypes.c_longlong
        L.sicl_or_decode.argtypes = [p, ctypes.c_size_t, p, ctypes.c_size_t, p]
        L.sicl_or_decode.restype = ctypes.c_longlong
        L.sicl_or_normalize.argtypes = [p, ctypes.c_uint32, p]
        L.sicl_or_normalize.restype = ctypes.c_int
        L.sicl_or_adler32.argtypes = [p, ctypes.c_size_t]
        L.sicl_or_adler32.restype = ctypes.c_uint32
        _LIB = L
    return _LIB


def codec_encode(latent: np.ndarray, image_wh=(0, 0), mode: int = 2, stream_symbols: int | None = None) -> bytes:
    """Oracle statement of the "SICL" container (sicn_codec_oracle.c). latent: [h][w][c] uint8.  stream_symbols (mode 3 only):
    the encoder's stream length, a power of two in 1024 .. 16384 (None = the default, 16384)."""
    latent = np.ascontiguousarray(latent, dtype=np.uint8)
    h, w, c = latent.shape
    L = lib()
    if stream_symbols is None:
        cap = L.sicl_or_max_bytes(mode, latent.size)
        out = np.zeros(max(cap, 64), np.uint8)
        n = L.sicl_or_encode(mode, _ptr(latent), w, h, c, image_wh[0], image_wh[1], _ptr(out), out.size)
    else:
        cap = L.sicl_or_max_bytes_sl(mode, latent.size, stream_symbols)
        out = np.zeros(max(cap, 64), np.uint8)
        n = L.sicl_or_encode_sl(mode, _ptr(latent), w, h, c, image_wh[0], image_wh[1], _ptr(out), out.size, stream_symbols)
    if n < 0:
        raise RuntimeError(f"sicl_or_encode rc={n}")
    return out[:n].tobytes()


def codec_decode(container: bytes):
    """Returns (latent [h][w][c], info[8] = mode, img_w, img_h, lat_w, lat_h, lat_c, n, payload)."""
    buf = np.frombuffer(container, np.uint8).copy()
    info = np.zeros(8, np.uint32)
    L = lib()
    probe = L.sicl_or_decode(_ptr(buf), buf.size, None, 0, _ptr(info))
    if probe not in (-28, 0):
        raise RuntimeError(f"sicl_or_decode rc={probe}")
    lat = np.zeros(max(int(info[6]), 1), np.uint8)
    n = L.sicl_or_decode(_ptr(buf), buf.size, _ptr(lat), lat.size, _ptr(info))
    if n < 0:
        raise RuntimeError(f"sicl_or_decode rc={n}")
    return lat[: int(info[6])].reshape(int(info[4]), int(info[3]), int(info[5])), info


def ctx_encode(latent: np.ndarray, scale: np.ndarray, image_wh=(0, 0)) -> bytes:
    """Container mode 4 (rANS-WC: 16 class tables, checkerboard context; oracle/sicn_hyper_oracle.c). [h][w][c] uint8 each."""
    latent = np.ascontiguousarray(latent, dtype=np.uint8)
    scale = np.ascontiguousarray(scale, dtype=np.uint8)
    assert latent.shape == scale.shape and latent.ndim == 3
    h, w, c = latent.shape
    L = lib()
    out = np.zeros(max(L.sicl_or_ctx_max_bytes(w, h, c), 64), np.uint8)
    n = L.sicl_or_ctx_encode(_ptr(latent), _ptr(scale), w, h, c, image_wh[0], image_wh[1], _ptr(out), out.size)
    if n < 0:
        raise RuntimeError(f"sicl_or_ctx_encode rc={n}")
    return out[:n].tobytes()


def ctx_decode(container: bytes, scale: np.ndarray):
    """Returns (latent [h][w][c], info[8]); `scale` must have the latent's shape."""
    buf = np.frombuffer(container, np.uint8).copy()
    scale = np.ascontiguousarray(scale, dtype=np.uint8)
    info = np.zeros(8, np.uint32)
    lat = np.zeros(max(scale.size, 1), np.uint8)
    n = lib().sicl_or_ctx_decode(_ptr(buf), buf.size, _ptr(scale), _ptr(lat), scale.size, _ptr(info))
    if n < 0:
        raise RuntimeError(f"sicl_or_ctx_decode rc={n}")
    return lat[: int(info[6])].reshape(int(info[4]), int(info[3]), int(info[5])), info


def _desc(d) -> OrDesc:
    return OrDesc(**{n: int(getattr(d, n)) for n, _ in OrDesc._fields_})


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def run_layer(d, words: np.ndarray, bias: np.ndarray, x: np.ndarray, form: str = "dataflow",
              threads: int = 1) -> np.ndarray:
    """Run one layer through the C oracle. form: 'dataflow' (FSM sliding window), 'dataflow_im2col',
    'naive' (the reference testbench's golden model) or 'direct' (closed form)."""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    bias = np.ascontiguousarray(bias, dtype=np.int8)
    x = np.ascontiguousarray(x, dtype=np.uint8)
    assert x.shape == (d.IFM_COL, d.IFM_ROW, d.IFM_CH), (x.shape, d)
    assert words.shape == (d.PE, d.W_TILES) and bias.shape == (d.OFM_CH,)
    out = np.empty((d.OFM_COL, d.OFM_ROW, d.OFM_CH), dtype=np.uint8)
    cd = _desc(d)
    L = lib()
    if form in ("dataflow", "dataflow_im2col"):
        fn = L.sicn_or_deconv522_dataflow if d.transposed else L.sicn_or_conv2d_dataflow
        rc = fn(ctypes.byref(cd), _ptr(words), _ptr(bias), _ptr(x), _ptr(out), int(form == "dataflow"))
    elif form == "naive":
        fn = L.sicn_or_naive_deconv2d if d.transposed else L.sicn_or_naive_conv2d
        rc = fn(ctypes.byref(cd), _ptr(words), _ptr(bias), _ptr(x), _ptr(out))
    elif form == "direct":
        rc = L.sicn_or_layer_direct(ctypes.byref(cd), _ptr(words), _ptr(bias), _ptr(x), _ptr(out), threads)
    else:
        raise ValueError(form)
    if rc != 0:
        raise RuntimeError(f"oracle {form} failed rc={rc}")
    return out


def run_layer_preact(d, words: np.ndarray, bias: np.ndarray, x: np.ndarray, threads: int = 1) -> np.ndarray:
    """The layer's lanes BEFORE the sign-bit ReLU (conv_nonsquare_top.cpp:272 without :273-275): input of the GDN extension."""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    bias = np.ascontiguousarray(bias, dtype=np.int8)
    x = np.ascontiguousarray(x, dtype=np.uint8)
    assert x.shape == (d.IFM_COL, d.IFM_ROW, d.IFM_CH), (x.shape, d)
    out = np.empty((d.OFM_COL, d.OFM_ROW, d.OFM_CH), dtype=np.uint8)
    cd = _desc(d)
    rc = lib().sicn_or_layer_direct_act(ctypes.byref(cd), _ptr(words), _ptr(bias), _ptr(x), _ptr(out), threads, 0)
    if rc != 0:
        raise RuntimeError(f"oracle direct_act failed rc={rc}")
    return out


def gdn(lanes: np.ndarray, beta: np.ndarray, gamma: np.ndarray, inverse: bool, shift: int) -> np.ndarray:
    """oracle/sicn_gdn_oracle.c: fixed-point GDN / IGDN over [...][C] pre-activation lanes (parity unpinned)."""
    lanes = np.ascontiguousarray(lanes, dtype=np.uint8)
    c = lanes.shape[-1]
    beta = np.ascontiguousarray(beta, dtype=np.uint32)
    gamma = np.ascontiguousarray(gamma, dtype=np.uint8)
    assert beta.shape == (c,) and gamma.shape == (c, c)
    out = np.empty_like(lanes)
    rc = lib().sicn_or_gdn(_ptr(lanes), _ptr(out), lanes.size // c, c, int(bool(inverse)), int(shift), _ptr(beta), _ptr(gamma))
    if rc != 0:
        raise RuntimeError(f"sicn_or_gdn rc={rc}")
    return out


def gdn_roots(n: np.ndarray, inverse: bool, shift: int) -> np.ndarray:
    """oracle/sicn_gdn_oracle.c, the root alone: r (binary32) for every n (uint32, >= 1)."""
    n = np.ascontiguousarray(n, dtype=np.uint32)
    r = np.empty(n.shape, np.float32)
    L = lib()
    L.sicn_or_gdn_roots.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
    L.sicn_or_gdn_roots.restype = ctypes.c_int
    rc = L.sicn_or_gdn_roots(_ptr(n), _ptr(r), n.size, int(bool(inverse)), int(shift))
    if rc != 0:
        raise RuntimeError(f"sicn_or_gdn_roots rc={rc}")
    return r


def gdn_outputs(x: np.ndarray, r: np.ndarray) -> np.ndarray:
    """oracle/sicn_gdn_oracle.c, the output step alone: the stored byte for lanes x (int8) and roots r (binary32)."""
    x = np.ascontiguousarray(x, dtype=np.int8)
    r = np.ascontiguousarray(r, dtype=np.float32)
    assert x.shape == r.shape
    y = np.empty(x.shape, np.uint8)
    L = lib()
    L.sicn_or_gdn_outputs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
    L.sicn_or_gdn_outputs.restype = ctypes.c_int
    L.sicn_or_gdn_outputs(_ptr(x), _ptr(r), _ptr(y), x.size)
    return y


def run_net(descs, words_list, bias_list, x: np.ndarray, form: str = "dataflow", threads: int = 1):
    outs = []
    for d, w, b in zip(descs, words_list, bias_list):
        x = run_layer(d, w, b, x, form, threads)
        outs.append(x)
    return outs



class OrConvLayerDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("K", "IFM_CH", "IFM_DIM", "OFM_CH", "OFM_DIM", "SIMD", "PE", "IN_BIT", "IN_SIGNED",
                                              "W_BIT", "W_TILES", "ACC_BIT", "ACC_SIGNED", "OUT_BIT", "activation", "NUM_TH",
                                              "ACT_VAL")]


def convlayer_dataflow(cdesc, words: np.ndarray, thresholds, x: np.ndarray, use_fsm: bool = True) -> np.ndarray:
    """ConvLayer_Batch as the dataflow runs it (sicn_or_convlayer_dataflow). cdesc: any object with the
    sicn_convlayer_desc fields. Returns uint32 [OFM_DIM][OFM_DIM][OFM_CH] (low OUT_BIT bits of each lane)."""
    d = OrConvLayerDesc(**{n: int(getattr(cdesc, n)) for n, _ in OrConvLayerDesc._fields_})
    L = lib()
    L.sicn_or_convlayer_dataflow.restype = ctypes.c_int
    L.sicn_or_convlayer_dataflow.argtypes = [ctypes.POINTER(OrConvLayerDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_int]
    words = np.ascontiguousarray(words, dtype=np.uint64)
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.zeros((d.OFM_DIM, d.OFM_DIM, d.OFM_CH), np.uint32)
    th = np.ascontiguousarray(thresholds, dtype=np.int32) if thresholds is not None else None
    rc = L.sicn_or_convlayer_dataflow(ctypes.byref(d), _ptr(words), _ptr(th) if th is not None else None, _ptr(x), _ptr(out),
                                      int(use_fsm))
    if rc:
        raise RuntimeError(f"sicn_or_convlayer_dataflow rc={rc}")
    return out
