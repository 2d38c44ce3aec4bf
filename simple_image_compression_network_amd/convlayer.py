"""`ConvLayer_Batch` — the generic finn-hlslib convolution layer (convlayer.h:89-125) — over the C ABI of
include/sicn_convlayer.h.  Same argument order as the reference:

    ConvLayer_Batch<K, IFMChannels, IFMDim, OFMChannels, OFMDim, SIMD, PE, TSrcI, TDstI, TWeightI>
                   (in, out, weights, activation, reps, r)

The template parameters and interpretation functors travel in `ConvLayerDesc`; `activation` is a
`PassThroughActivation` or a `ThresholdsActivation` object.  Parity status: unpinned (the reference never
instantiates this surface: conv_nonsquare_top.cpp:223 is commented out)."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, fields

import numpy as np

from . import _lib
from .api import FixedPointWeights, _stream_ptr

__all__ = ["ConvLayerDesc", "PassThroughActivation", "ThresholdsActivation", "ConvLayer", "ConvLayer_Batch"]


@dataclass(frozen=True)
class PassThroughActivation:
    """activations.hpp:127-134 — `PassThroughActivation<ap_int<ACC_BIT>>` / `<ap_uint<ACC_BIT>>`."""
    ACC_BIT: int = 16
    ACC_SIGNED: bool = True


@dataclass(frozen=True)
class ThresholdsActivation:
    """activations.hpp:168-190 — `ThresholdsActivation<NF, PE, NumTH, TA, TR, ActVal>`;
    `m_thresholds` is [PE][NF][NumTH] int32, result = ActVal + #{i : m_thresholds[pe][nf][i] < accu}."""
    m_thresholds: np.ndarray
    ACC_BIT: int = 16
    ACC_SIGNED: bool = True
    ACT_VAL: int = 0


@dataclass(frozen=True)
class ConvLayerDesc:
    K: int
    IFM_CH: int
    IFM_DIM: int
    OFM_CH: int
    SIMD: int
    PE: int
    W_BIT: int = 4
    IN_SIGNED: bool = False
    OUT_BIT: int = 8
    IN_BIT: int = 8       # TSrcI::width: 1, 2, 4 or 8 — the input stream carries IFM_CH * IN_BIT bits per pixel (convlayer.h:100)

    @property
    def OFM_DIM(self) -> int:
        return self.IFM_DIM - self.K + 1

    @property
    def W_TILES(self) -> int:
        return (self.OFM_CH // self.PE) * (self.K * self.K * self.IFM_CH // self.SIMD)

    def to_c(self, activation) -> "_lib.CConvLayerDesc":
        th = isinstance(activation, ThresholdsActivation)
        return _lib.CConvLayerDesc(K=self.K, IFM_CH=self.IFM_CH, IFM_DIM=self.IFM_DIM, OFM_CH=self.OFM_CH, OFM_DIM=self.OFM_DIM,
                                   SIMD=self.SIMD, PE=self.PE, IN_BIT=self.IN_BIT, IN_SIGNED=int(self.IN_SIGNED), W_BIT=self.W_BIT,
                                   W_TILES=self.W_TILES, ACC_BIT=activation.ACC_BIT, ACC_SIGNED=int(activation.ACC_SIGNED),
                                   OUT_BIT=self.OUT_BIT, activation=int(th),
                                   NUM_TH=int(activation.m_thresholds.shape[2]) if th else 0,
                                   ACT_VAL=int(activation.ACT_VAL) if th else 0)


class ConvLayer:
    """A layer with its parameters resident on the device: descriptor, weights and activation are validated and uploaded ONCE (the reference's
    weights and thresholds are objects the caller builds once and passes by reference, convlayer.h:89-111), every call only enqueues.
    `layer(in_, out=None, reps=1, stream=None, kernel=0)` — tensors as for ConvLayer_Batch.  `close()` (or garbage collection) frees the handle;
    the caller keeps it alive until the last enqueued call has run."""

    def __init__(self, desc: ConvLayerDesc, weights: FixedPointWeights, activation):
        L = _lib.lib()
        self.desc, self.activation = desc, activation
        self._cd = desc.to_c(activation)
        _lib.check(L.sicn_convlayer_validate(ctypes.byref(self._cd)), "sicn_convlayer_validate")
        words = np.ascontiguousarray(weights.m_weights, dtype=np.uint64)
        if words.shape != (desc.PE, desc.W_TILES):
            raise ValueError("FixedPointWeights fold does not match the layer")
        thr = None
        if isinstance(activation, ThresholdsActivation):
            thr = np.ascontiguousarray(activation.m_thresholds, dtype=np.int32)
            if thr.shape[:2] != (desc.PE, desc.OFM_CH // desc.PE):
                raise ValueError("m_thresholds must be [PE][NF][NumTH]")
        self._h = ctypes.c_void_p()
        _lib.check(L.sicn_convlayer_params_create(ctypes.byref(self._cd), words.ctypes.data_as(ctypes.c_void_p), 8,
                                                  thr.ctypes.data_as(ctypes.c_void_p) if thr is not None else None,
                                                  ctypes.byref(self._h)), "sicn_convlayer_params_create")

    def shapes(self, reps: int):
        import torch
        d = self.desc
        dt = {2: torch.uint8, 4: torch.uint8, 8: torch.uint8, 16: torch.int16, 32: torch.int32}[d.OUT_BIT]
        return ((reps, d.IFM_DIM, d.IFM_DIM, d.IFM_CH * d.IN_BIT // 8),
                (reps, d.OFM_DIM, d.OFM_DIM, d.OFM_CH * d.OUT_BIT // 8 if d.OUT_BIT < 8 else d.OFM_CH), dt)

    def __call__(self, in_, out=None, reps: int = 1, stream=None, kernel: int = 0):
        import torch
        if not self._h:
            raise RuntimeError("ConvLayer is closed")
        shape_in, shape_out, dt = self.shapes(reps)
        if not (in_.is_cuda and in_.dtype == torch.uint8 and in_.is_contiguous() and tuple(in_.shape) == shape_in):
            raise TypeError(f"in: need a contiguous CUDA uint8 tensor of shape {shape_in}")
        if out is None:
            out = torch.empty(shape_out, dtype=dt, device=in_.device)
        if not (out.is_cuda and out.dtype == dt and out.is_contiguous() and tuple(out.shape) == shape_out):
            raise TypeError(f"out: need a contiguous CUDA {dt} tensor of shape {shape_out}")
        _lib.check(_lib.lib().sicn_conv_layer_batch_kernel(ctypes.byref(self._cd), self._h, ctypes.c_void_p(in_.data_ptr()),
                                                           ctypes.c_void_p(out.data_ptr()), reps, int(kernel), _stream_ptr(stream)),
                   "sicn_conv_layer_batch_kernel")
        return out

    def close(self):
        if getattr(self, "_h", None):
            try:
                if _lib._lib is not None:
                    _lib._lib.sicn_convlayer_params_free(self._h)
            except Exception:   # interpreter shutdown
                pass
            self._h = None

    __del__ = close


def ConvLayer_Batch(desc: ConvLayerDesc, in_, out, weights: FixedPointWeights, activation, reps: int = 1, stream=None,
                    kernel: int = 0):
    """in_: CUDA uint8 [reps][IFM_DIM][IFM_DIM][IFM_CH * IN_BIT / 8] — the stream words of the pixels, lane c in bits [c IN_BIT, (c + 1) IN_BIT)
    (one byte per lane at IN_BIT = 8); out: CUDA tensor [reps][OFM_DIM][OFM_DIM][OFM_CH] of dtype uint8 / int16 / int32 matching OUT_BIT
    (8 / 16 / 32), or uint8 [reps][OFM_DIM][OFM_DIM][OFM_CH * OUT_BIT / 8] for OUT_BIT 2 / 4 (packed exactly like an input stream, so it
    can be handed to the next layer as it is), or None to allocate.  Returns `out`.
    kernel: 0 = automatic (MFMA kernel when the shape allows), 1 = the direct kernel (tests compare the two).
    The reference's calling convention: parameters travel with every call (uploaded, used, freed; the call returns when the layer has run).
    `ConvLayer` keeps them resident."""
    import torch
    layer = ConvLayer(desc, weights, activation)
    try:
        out = layer(in_, out, reps, stream, kernel)
        (stream if stream is not None and hasattr(stream, "synchronize") else torch.cuda.current_stream()).synchronize()   # the handle is freed below
    finally:
        layer.close()
    return out
