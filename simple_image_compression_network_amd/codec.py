"""Latent container ("SICL" v1) + static rANS coder on the GPU — Python wrapper of include/sicn_codec.h.

EXTENSION BEYOND THE REFERENCE: the reference has no coder, so nothing here
mirrors a reference interface and parity is "unpinned" (own specification: oracle/sicn_codec_oracle.c)."""
from __future__ import annotations

import ctypes

from . import _lib

RAW8, PACKED7, RANS, RANSW = 0, 1, 2, 3
__all__ = ["RAW8", "PACKED7", "RANS", "RANSW", "encode_latent", "decode_latent", "parse_header"]


def _stream_ptr(stream):
    import torch
    stream = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(getattr(stream, "cuda_stream", stream))


def encode_latent(latent, image_width: int, image_height: int, mode: int = RANSW, stream=None):
    """latent: CUDA uint8 tensor [H/16][W/16][C] (one image).  Returns a CUDA uint8 tensor holding the container."""
    import torch
    L = _lib.lib()
    if not (latent.is_cuda and latent.dtype == torch.uint8 and latent.is_contiguous() and latent.dim() == 3):
        raise TypeError("latent must be a contiguous CUDA uint8 tensor [h][w][c]")
    h, w, c = (int(v) for v in latent.shape)
    n = h * w * c
    out = torch.empty(max(L.sicn_codec_max_bytes(mode, n), 64), dtype=torch.uint8, device=latent.device)
    ws = torch.empty(max(L.sicn_codec_workspace_bytes(mode, n), 64), dtype=torch.uint8, device=latent.device)
    nbytes = ctypes.c_size_t(0)
    _lib.check(L.sicn_codec_encode(mode, ctypes.c_void_p(latent.data_ptr()), w, h, c, image_width, image_height,
                                   ctypes.c_void_p(out.data_ptr()), out.numel(), ctypes.byref(nbytes),
                                   ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr(stream)), "sicn_codec_encode")
    return out[: nbytes.value]


def parse_header(container_bytes: bytes) -> "_lib.CodecInfo":
    info = _lib.CodecInfo()
    buf = (ctypes.c_uint8 * len(container_bytes)).from_buffer_copy(container_bytes)
    _lib.check(_lib.lib().sicn_codec_parse_header(buf, len(container_bytes), ctypes.byref(info)), "sicn_codec_parse_header")
    return info


def decode_latent(container, stream=None):
    """container: CUDA uint8 tensor.  Returns (latent [h][w][c] CUDA uint8, CodecInfo)."""
    import torch
    L = _lib.lib()
    info = parse_header(bytes(container[:48].cpu().numpy().tobytes()))
    n = int(info.n_symbols)
    latent = torch.empty((int(info.lat_h), int(info.lat_w), int(info.lat_c)), dtype=torch.uint8, device=container.device)
    ws = torch.empty(max(L.sicn_codec_workspace_bytes(int(info.mode), n), 64), dtype=torch.uint8, device=container.device)
    container = container.contiguous()
    _lib.check(L.sicn_codec_decode(ctypes.c_void_p(container.data_ptr()), container.numel(),
                                   ctypes.c_void_p(latent.data_ptr()), max(n, 1), ctypes.byref(info),
                                   ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr(stream)), "sicn_codec_decode")
    return latent, info
