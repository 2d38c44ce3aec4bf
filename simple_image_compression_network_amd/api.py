"""Host-side mirror of the reference's operator interface for the transform path.

Reference surface (conv_nonsquare_top.cpp):
    conv2d<...>(weights, bias, in, out, numReps)            :198-280
    deconv522<...>(weights, bias, in, out, numReps)         :71-195
    conv2d_layer0(in, out, numReps)                         :282-286
    deconv2d_layer4(in, out, numReps)                       :288-291
    eight_layers_net(in, out, numReps)                      :295-357
    FixedPointWeights<SIMD, ap_int<4>, PE, TILES>           weights.hpp:110-150

Same names, argument order and meaning; the `hls::stream<ap_uint<C*8>>` arguments become CUDA/HIP
`torch.uint8` tensors of shape [numReps][H][W][C] (byte-identical to the stream), and
`numReps` is a true batch (the reference is only well defined at numReps == 1).
Everything here is plumbing over the C ABI of include/sicn.h: torch supplies device memory and
streams, nothing else.  No CPU path exists: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes
from dataclasses import replace
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .config import CLayerDesc, LayerDesc, REFERENCE_DESCS, eight_layer_descs

__all__ = ["FixedPointWeights", "DeviceWeights", "conv2d", "deconv522", "conv2d_layer0", "deconv2d_layer4",
           "eight_layers_net", "EightLayersNet", "load_param_weights", "PARAM", "GDN"]

_DATA = Path(__file__).resolve().parent / "data" / "param_weights.npz"


class FixedPointWeights:
    """`FixedPointWeights<SIMD, ap_int<W_BIT>, PE, TILES>` (weights.hpp:110-150): `m_weights[PE][TILES]`
    words, element s of a word = sign-extended nibble in bits [4s, 4s+4)."""

    def __init__(self, SIMD: int, W_BIT: int, PE: int, TILES: int, m_weights):
        if not 2 <= W_BIT <= 8 or SIMD * W_BIT > 64:
            raise ValueError("W_BIT must be 2..8 and SIMD*W_BIT <= 64 (the net itself uses ap_int<4> tiles, ap_int<8> biases)")
        self.SIMD, self.W_BIT, self.PE, self.TILES = SIMD, W_BIT, PE, TILES
        self.m_weights = np.ascontiguousarray(m_weights, dtype=np.uint64).reshape(PE, TILES)

    def bias_values(self) -> np.ndarray:
        """`bias.weights(j)[0][0]` for every j (conv_nonsquare_top.cpp:272) as int8."""
        assert self.SIMD == 1 and self.PE == 1 and self.W_BIT == 8
        return self.m_weights.reshape(-1).astype(np.uint8).view(np.int8)


def _stream_ptr(stream) -> ctypes.c_void_p:
    if stream is None:
        import torch
        stream = torch.cuda.current_stream()
    return ctypes.c_void_p(getattr(stream, "cuda_stream", stream))


def _check_tensor(t, shape, what):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
        raise TypeError(f"{what}: need a contiguous torch.uint8 CUDA tensor (no CPU path exists)")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{what}: shape {tuple(t.shape)} != {tuple(shape)}")


class DeviceWeights:
    """One layer's weights + bias uploaded through `sicn_weights_from_finn_tiles`."""

    def __init__(self, desc: LayerDesc, weights: FixedPointWeights, bias):
        L = _lib.lib()
        if (weights.SIMD, weights.PE, weights.TILES) != (desc.SIMD, desc.PE, desc.W_TILES):
            raise ValueError("FixedPointWeights fold does not match the layer descriptor")
        b = bias.bias_values() if isinstance(bias, FixedPointWeights) else np.ascontiguousarray(bias, np.int8)
        if b.shape != (desc.OFM_CH,):
            raise ValueError("bias must have OFM_CH entries")
        self._h = ctypes.c_void_p()
        cd = desc.to_c()
        _lib.check(L.sicn_weights_from_finn_tiles(ctypes.byref(cd), weights.m_weights.ctypes.data_as(ctypes.c_void_p),
                                                  8, b.ctypes.data_as(ctypes.c_void_p), ctypes.byref(self._h)),
                   "sicn_weights_from_finn_tiles")
        self.key = (desc.IFM_CH, desc.OFM_CH, desc.transposed)

    @property
    def handle(self) -> ctypes.c_void_p:
        return self._h

    def __del__(self):
        try:
            if getattr(self, "_h", None) and _lib._lib is not None:
                _lib._lib.sicn_weights_free(self._h)
                self._h = None
        except Exception:       # interpreter shutdown: module globals may already be gone
            pass


class GDN:
    """Fixed-point GDN (inverse=False) / IGDN (inverse=True) activation, include/sicn_gdn.h.  EXTENSION BEYOND THE
    REFERENCE (it has no GDN, activations.hpp:127-224); replaces a layer's sign-bit ReLU (conv_nonsquare_top.cpp:273-275).
    beta: uint32 [C] in [1, 65535]; gamma: uint8 [C][C] in [0, 127]; shift in [1, 24]."""

    def __init__(self, beta, gamma, inverse: bool = False, shift: int = 12):
        beta = np.ascontiguousarray(beta, dtype=np.uint32)
        gamma = np.ascontiguousarray(gamma, dtype=np.uint8)
        c = int(beta.shape[0])
        if gamma.shape != (c, c):
            raise ValueError("gamma must be [C][C]")
        self.channels, self.inverse, self.shift = c, bool(inverse), int(shift)
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().sicn_gdn_create(c, int(self.inverse), self.shift, beta.ctypes.data_as(ctypes.c_void_p),
                                              gamma.ctypes.data_as(ctypes.c_void_p), ctypes.byref(self._h)), "sicn_gdn_create")

    @property
    def handle(self) -> ctypes.c_void_p:
        return self._h

    def apply_(self, lanes, stream=None):
        """In place over a contiguous CUDA uint8 tensor [...][C] of pre-activation lanes."""
        import torch
        if not (isinstance(lanes, torch.Tensor) and lanes.is_cuda and lanes.dtype == torch.uint8 and lanes.is_contiguous()
                and lanes.shape[-1] == self.channels):
            raise TypeError("lanes: need a contiguous CUDA uint8 tensor [...][C]")
        _lib.check(_lib.lib().sicn_gdn_apply(self._h, ctypes.c_void_p(lanes.data_ptr()), lanes.numel() // self.channels,
                                             _stream_ptr(stream)), "sicn_gdn_apply")
        return lanes

    def __del__(self):
        try:
            if getattr(self, "_h", None) and _lib._lib is not None:
                _lib._lib.sicn_gdn_free(self._h)
                self._h = None
        except Exception:       # interpreter shutdown
            pass


def _as_device(desc, weights, bias) -> DeviceWeights:
    return weights if isinstance(weights, DeviceWeights) else DeviceWeights(desc, weights, bias)


def _opt_ptr(options):
    """options: None (library defaults), a dict of sicn_options fields, or a _lib.COptions."""
    if options is None:
        return None
    if isinstance(options, dict):
        options = _lib.make_options(**options)
    return ctypes.byref(options)


def _run(fn_name, desc, weights, bias, in_, out, numReps, stream, options=None, gdn=None):
    import torch
    L = _lib.lib()
    desc.validate()
    dw = _as_device(desc, weights, bias)
    _check_tensor(in_, (numReps,) + desc.in_shape, "in")
    if out is None:
        out = torch.empty((numReps,) + desc.out_shape, dtype=torch.uint8, device=in_.device)
    _check_tensor(out, (numReps,) + desc.out_shape, "out")
    cd = desc.to_c()
    if gdn is not None:
        _lib.check(getattr(L, fn_name + "_gdn")(ctypes.byref(cd), dw.handle, gdn.handle, ctypes.c_void_p(in_.data_ptr()),
                                                ctypes.c_void_p(out.data_ptr()), numReps, _opt_ptr(options),
                                                _stream_ptr(stream)), fn_name + "_gdn")
        return out
    _lib.check(getattr(L, fn_name + "_opt")(ctypes.byref(cd), dw.handle, ctypes.c_void_p(in_.data_ptr()),
                                            ctypes.c_void_p(out.data_ptr()), numReps, _opt_ptr(options),
                                            _stream_ptr(stream)), fn_name)
    return out


def conv2d(desc: LayerDesc, weights, bias, in_, out=None, numReps: int = 1, stream=None, options=None, gdn=None):
    """`conv2d<...>(weights, bias, in, out, numReps)` — conv_nonsquare_top.cpp:198-280.
    `options`: sicn_options fields as a dict (kernel-selection knobs for tests / experiments).
    `gdn`: a GDN object to apply in place of the ReLU (extension beyond the reference)."""
    return _run("sicn_conv2d", desc, weights, bias, in_, out, numReps, stream, options, gdn)


def deconv522(desc: LayerDesc, weights, bias, in_, out=None, numReps: int = 1, stream=None, options=None, gdn=None):
    """`deconv522<...>(weights, bias, in, out, numReps)` — conv_nonsquare_top.cpp:71-195."""
    return _run("sicn_deconv522", desc, weights, bias, in_, out, numReps, stream, options, gdn)


# ---- the PARAM:: tables (memdata_nonsquare.h) -------------------------------------------------
def load_param_weights(path=None):
    """[(FixedPointWeights weights_layerN, FixedPointWeights bias_layerN)] * 8 — `namespace PARAM`."""
    z = np.load(path or _DATA)
    out = []
    for n in range(8):
        simd, wbit, pe, tiles = (int(v) for v in z[f"w{n}_meta"])
        b = z[f"b{n}"]
        out.append((FixedPointWeights(simd, wbit, pe, tiles, z[f"w{n}_words"]),
                    FixedPointWeights(1, 8, 1, b.size, b.view(np.uint8).astype(np.uint64))))
    return out


class _Param:
    """Lazy `PARAM::weights_layerN` / `PARAM::bias_layerN`."""
    _tables = None

    def __getattr__(self, name):
        if _Param._tables is None:
            _Param._tables = load_param_weights()
        kind, _, n = name.partition("_layer")
        if kind in ("weights", "bias") and n.isdigit() and int(n) < 8:
            return _Param._tables[int(n)][0 if kind == "weights" else 1]
        raise AttributeError(name)


PARAM = _Param()


class EightLayersNet:
    """`eight_layers_net` (conv_nonsquare_top.cpp:295-357) for one image size: descriptors, device
    weights, the chain handle and its ping-pong workspace.  `forward` enqueues the 8 layers on the
    current stream and returns (reconstruction, latent)."""

    def __init__(self, width: int = 768, height: int = 512, params=None, device=None,
                 descs: Optional[Sequence[LayerDesc]] = None, shared_weights: Optional[Sequence[DeviceWeights]] = None,
                 options=None, gdn: Optional[Sequence[Optional["GDN"]]] = None):
        import torch
        L = _lib.lib()
        self.descs: List[LayerDesc] = list(descs) if descs is not None else eight_layer_descs(width, height)
        self.device = torch.device(device if device is not None else "cuda")
        if shared_weights is not None:
            self.weights = list(shared_weights)
        else:
            params = params if params is not None else load_param_weights()
            with torch.cuda.device(self.device):
                self.weights = [DeviceWeights(d, w, b) for d, (w, b) in zip(self.descs, params)]
        n = len(self.descs)
        cdescs = (CLayerDesc * n)(*[d.to_c() for d in self.descs])
        handles = (ctypes.c_void_p * n)(*[w.handle for w in self.weights])
        self._h = ctypes.c_void_p()
        self.gdn = list(gdn) if gdn is not None else None      # keeps the activations alive
        if self.gdn is not None:
            if len(self.gdn) != n:
                raise ValueError("gdn must have one entry (or None) per layer")
            ghandles = (ctypes.c_void_p * n)(*[(g.handle if g is not None else None) for g in self.gdn])
            _lib.check(L.sicn_net_create_gdn(cdescs, handles, ghandles, n, _opt_ptr(options), ctypes.byref(self._h)),
                       "sicn_net_create_gdn")
        else:
            _lib.check(L.sicn_net_create_opt(cdescs, handles, n, _opt_ptr(options), ctypes.byref(self._h)), "sicn_net_create_opt")
        self._ws = None
        self._ws_bytes = {}

    def __del__(self):
        try:
            if getattr(self, "_h", None) and _lib._lib is not None:
                _lib._lib.sicn_net_free(self._h)
                self._h = None
        except Exception:       # interpreter shutdown
            pass

    def workspace(self, n_images: int):
        import torch
        # sized per batch size (the K-split scratch of a small batch may exceed what a larger one asks for); never shrinks.
        # No initialisation is needed (sicn.h: the arrival words carry the net's random tag).
        nbytes = self._ws_bytes.get(n_images)
        if nbytes is None:
            nbytes = self._ws_bytes[n_images] = max(int(_lib.lib().sicn_net_workspace_bytes(self._h, n_images)), 256)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward(self, in_, out=None, latent=None, numReps: Optional[int] = None, want_latent: bool = True, stream=None):
        import torch
        L = _lib.lib()
        n = in_.shape[0] if numReps is None else numReps
        _check_tensor(in_, (n,) + self.descs[0].in_shape, "in")
        if out is None:
            out = torch.empty((n,) + self.descs[-1].out_shape, dtype=torch.uint8, device=in_.device)
        _check_tensor(out, (n,) + self.descs[-1].out_shape, "out")
        lat_ptr = ctypes.c_void_p(0)
        if want_latent and len(self.descs) > 3:
            if latent is None:
                latent = torch.empty((n,) + self.descs[3].out_shape, dtype=torch.uint8, device=in_.device)
            _check_tensor(latent, (n,) + self.descs[3].out_shape, "latent")
            lat_ptr = ctypes.c_void_p(latent.data_ptr())
        ws = self.workspace(n)
        _lib.check(L.sicn_eight_layers_net(self._h, ctypes.c_void_p(in_.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                           lat_ptr, n, ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr(stream)),
                   "sicn_eight_layers_net")
        return out, latent

    def capture(self, in_, out, latent=None):
        """The whole forward pass as ONE replayable hipGraph (the launch functions neither allocate nor
        synchronise, sicn.h): for small images the 8 launches cost more host time than device time.
        Returns a torch.cuda.CUDAGraph; `.replay()` recomputes `out` / `latent` from the current contents
        of `in_` (same tensors).  Keep other frees (net objects, tensors) out of the capture window."""
        import gc
        import torch
        self.forward(in_, out, latent, want_latent=latent is not None)      # warm-up: module load, workspace
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=in_.device)
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    self.forward(in_, out, latent, want_latent=latent is not None)
        finally:
            if gc_was:
                gc.enable()
        return graph

    def analysis(self, in_, latent=None, stream=None):
        """Encoder half: layers 0..3, image batch -> latent (conv_3_out, conv_nonsquare_top.cpp:322-325)."""
        return self.run_layers(0, 3, in_, out=latent, stream=stream)[0]

    def synthesis(self, latent, out=None, stream=None):
        """Decoder half: layers 4..7, latent -> reconstruction."""
        return self.run_layers(4, len(self.descs) - 1, latent, out=out, stream=stream)[0]

    def run_layers(self, first: int, last: int, in_, tap_layer: int = -1, stream=None, out=None):
        """Layers [first, last] of the chain (sicn_net_forward); returns (out, tap or None)."""
        import torch
        L = _lib.lib()
        n = in_.shape[0]
        _check_tensor(in_, (n,) + self.descs[first].in_shape, "in")
        if out is None:
            out = torch.empty((n,) + self.descs[last].out_shape, dtype=torch.uint8, device=in_.device)
        _check_tensor(out, (n,) + self.descs[last].out_shape, "out")
        tap = None
        tap_ptr = ctypes.c_void_p(0)
        if tap_layer >= 0:
            tap = torch.empty((n,) + self.descs[tap_layer].out_shape, dtype=torch.uint8, device=in_.device)
            tap_ptr = ctypes.c_void_p(tap.data_ptr())
        ws = self.workspace(n)
        _lib.check(L.sicn_net_forward(self._h, first, last, ctypes.c_void_p(in_.data_ptr()),
                                      ctypes.c_void_p(out.data_ptr()), tap_layer, tap_ptr, n,
                                      ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr(stream)), "sicn_net_forward")
        return out, tap

    # measurement aids (sicn_net_profile / sicn_net_layer_ms)
    def profile(self, enable: bool = True) -> None:
        _lib.check(_lib.lib().sicn_net_profile(self._h, int(enable)), "sicn_net_profile")

    def layer_ms(self, reset: bool = True):
        n = len(self.descs)
        ms = (ctypes.c_float * n)()
        cnt = (ctypes.c_int * n)()
        _lib.check(_lib.lib().sicn_net_layer_ms(self._h, int(reset), ms, cnt), "sicn_net_layer_ms")
        return list(ms), list(cnt)


_DEFAULT_NETS = {}


def _default_net(width, height, device) -> EightLayersNet:
    key = (width, height, str(device))
    if key not in _DEFAULT_NETS:
        _DEFAULT_NETS[key] = EightLayersNet(width, height, device=device)
    return _DEFAULT_NETS[key]


def conv2d_layer0(in_, out=None, numReps: int = 1, stream=None):
    """`conv2d_layer0(in, out, numReps)` — conv_nonsquare_top.cpp:282-286: layer 0 with
    PARAM::weights_layer0 / bias_layer0; image size taken from `in_` ([numReps][H][W][3])."""
    h, w = int(in_.shape[1]), int(in_.shape[2])
    net = _default_net(w, h, in_.device)
    return conv2d(net.descs[0], net.weights[0], None, in_, out, numReps, stream)


def deconv2d_layer4(in_, out=None, numReps: int = 1, stream=None):
    """`deconv2d_layer4(in, out, numReps)` — conv_nonsquare_top.cpp:288-291 (PARAM layer 4)."""
    h, w = int(in_.shape[1]), int(in_.shape[2])
    d = replace(REFERENCE_DESCS[4], IFM_ROW=w, IFM_COL=h, OFM_ROW=2 * w, OFM_COL=2 * h)
    net = _default_net(16 * w, 16 * h, in_.device)
    return deconv522(d, net.weights[4], None, in_, out, numReps, stream)


def eight_layers_net(in_, out=None, numReps: int = 1, stream=None):
    """`eight_layers_net(in, out, numReps)` — conv_nonsquare_top.cpp:295-357 with the PARAM tables."""
    h, w = int(in_.shape[1]), int(in_.shape[2])
    net = _default_net(w, h, in_.device)
    return net.forward(in_, out, numReps=numReps, want_latent=False, stream=stream)[0]
