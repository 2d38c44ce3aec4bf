"""Latent container ("SICL" v1) + static rANS coder on the GPU — Python wrapper of include/sicn_codec.h.

EXTENSION BEYOND THE REFERENCE: the reference has no coder, so nothing here
mirrors a reference interface and parity is "unpinned" (own specification: oracle/sicn_codec_oracle.c)."""
from __future__ import annotations

import ctypes

from . import _lib

RAW8, PACKED7, RANS, RANSW, RANSWC = 0, 1, 2, 3, 4
__all__ = ["RAW8", "PACKED7", "RANS", "RANSW", "RANSWC", "encode_latent", "decode_latent", "parse_header", "encode_latents",
           "decode_latents", "LatentCoder", "ContextCoder", "auto_stream_symbols", "WSTREAM_SYMBOLS"]


WSTREAM_SYMBOLS = 16384   # SICN_CODEC_WSTREAM_SYMBOLS: the default (and longest) rANS-W stream


def auto_stream_symbols(n_symbols: int, n_images: int = 1) -> int:
    """Stream length of the rANS-W coder as a function of ONE image's latent only (`n_images` is accepted and ignored: the
    bitstream of an image must not depend on how many images were coded beside it — ADVICE r3; a container written by a batch of
    8 decodes in a batch of 1).  A stream is a serial chain on one wave and ends with 260 bytes of flush (its 64 final states and
    its length entry), so the choice trades latency against bytes:
      * 16384 (the format's default, 2.8 % flush at 4.5 bit / symbol) for latents of at least 128 such streams — a 4K image's
        6.2 M symbols are 380;
      * 8192 otherwise (one 1080p latent: 192 streams, half the chain, + 2.8 % bytes: 3.54 instead of 3.45 bit / pixel).
    Shorter streams remain an explicit choice (`stream_symbols=2048`: a quarter of that chain again, + 14 % bytes)."""
    del n_images
    return 16384 if n_symbols >= 128 * 16384 else 8192


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
    need = L.sicn_codec_workspace_bytes_sl(n, int(info.stream_symbols)) if int(info.mode) == RANSW else L.sicn_codec_workspace_bytes(int(info.mode), n)
    ws = torch.empty(max(need, 64), dtype=torch.uint8, device=container.device)
    container = container.contiguous()
    _lib.check(L.sicn_codec_decode(ctypes.c_void_p(container.data_ptr()), container.numel(),
                                   ctypes.c_void_p(latent.data_ptr()), max(n, 1), ctypes.byref(info),
                                   ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr(stream)), "sicn_codec_decode")
    return latent, info


def encode_latents(latents, image_width: int, image_height: int, stream=None):
    """latents: CUDA uint8 tensor [n][H/16][W/16][C].  One rANS-W container per image, byte-identical to
    `encode_latent(latents[i], ...)`, with two host synchronisations for the whole batch.  Returns
    (slots, sizes): a CUDA uint8 tensor [n][slot_bytes] and the list of valid byte counts."""
    import torch
    L = _lib.lib()
    if not (latents.is_cuda and latents.dtype == torch.uint8 and latents.is_contiguous() and latents.dim() == 4):
        raise TypeError("latents must be a contiguous CUDA uint8 tensor [n][h][w][c]")
    nimg, h, w, c = (int(v) for v in latents.shape)
    n = h * w * c
    slot = (int(L.sicn_codec_max_bytes(RANSW, n)) + 255) // 256 * 256
    out = torch.empty((nimg, slot), dtype=torch.uint8, device=latents.device)
    ws = torch.empty(max(L.sicn_codec_batch_workspace_bytes(RANSW, n, nimg), 64), dtype=torch.uint8, device=latents.device)
    sizes = (ctypes.c_size_t * max(nimg, 1))()
    _lib.check(L.sicn_codec_encode_batch(RANSW, ctypes.c_void_p(latents.data_ptr()), nimg, w, h, c, image_width, image_height,
                                         ctypes.c_void_p(out.data_ptr()), slot, sizes, ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                         _stream_ptr(stream)), "sicn_codec_encode_batch")
    return out, [int(sizes[i]) for i in range(nimg)]


def decode_latents(slots, sizes, stream=None):
    """Inverse of `encode_latents`: (slots [n][slot_bytes], sizes) -> (latents [n][h][w][c], [CodecInfo])."""
    import torch
    L = _lib.lib()
    nimg, slot = (int(v) for v in slots.shape)
    info0 = parse_header(bytes(slots[0, :48].cpu().numpy().tobytes()))
    n = int(info0.n_symbols)
    latents = torch.empty((nimg, int(info0.lat_h), int(info0.lat_w), int(info0.lat_c)), dtype=torch.uint8, device=slots.device)
    ws = torch.empty(max(L.sicn_codec_batch_workspace_bytes_sl(n, nimg, int(info0.stream_symbols)), 64), dtype=torch.uint8, device=slots.device)
    csz = (ctypes.c_size_t * nimg)(*sizes)
    infos = (_lib.CodecInfo * nimg)()
    slots = slots.contiguous()
    _lib.check(L.sicn_codec_decode_batch(ctypes.c_void_p(slots.data_ptr()), slot, csz, nimg, ctypes.c_void_p(latents.data_ptr()),
                                         max(n, 1), infos, ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr(stream)),
               "sicn_codec_decode_batch")
    return latents, list(infos)


class LatentCoder:
    """The asynchronous rANS-W batch pair (sicn_codec_encode_batch_async / sicn_codec_decode_batch_async) with its
    buffers allocated once: n latents of one shape per call, nothing but kernel launches on the current stream — no
    host synchronisation, so analysis -> encode -> decode -> synthesis can be enqueued back to back (or captured in
    one hipGraph).  Verdicts and sizes stay on the device until `check()` / `sizes()` fetch them."""

    def __init__(self, n_images: int, lat_h: int, lat_w: int, lat_c: int, image_width: int = 0, image_height: int = 0,
                 device="cuda", stream_symbols=None):
        """stream_symbols: None or "auto" = `auto_stream_symbols` of ONE image's latent (16384 for large latents, 8192 below 2 M
        symbols), or a power of two 1024 .. 16384 (sicn_codec_*_async_sl: shorter streams = shorter critical path on small latents,
        260 bytes per extra stream).  A decoder object must be built with the encoder's value — the container header carries it:
        `LatentCoder.for_containers(slots, n)` builds the matching decoder for containers made elsewhere (e.g. by the C entry points
        without `_sl`, which always write 16384-symbol streams, or by another stream length on the encoder's side)."""
        import torch
        L = _lib.lib()
        self.shape = (int(n_images), int(lat_h), int(lat_w), int(lat_c))
        self.image_wh = (int(image_width), int(image_height))
        n = lat_h * lat_w * lat_c
        self.stream_symbols = auto_stream_symbols(n) if stream_symbols in (None, "auto") else int(stream_symbols)
        if self.stream_symbols not in (1024, 2048, 4096, 8192, 16384):
            raise ValueError("stream_symbols must be a power of two in 1024 .. 16384")
        self.slot = (int(L.sicn_codec_max_bytes_sl(n, self.stream_symbols)) + 255) // 256 * 256
        dev = torch.device(device)
        self.slots = torch.empty((n_images, self.slot), dtype=torch.uint8, device=dev)
        self.ws = torch.empty(max(L.sicn_codec_batch_workspace_bytes_sl(n, n_images, self.stream_symbols), 256), dtype=torch.uint8, device=dev)
        self.enc_status = torch.zeros((n_images, 2), dtype=torch.int32, device=dev)   # sicn_codec_status {error, bytes}
        self.dec_status = torch.zeros((n_images, 2), dtype=torch.int32, device=dev)

    @classmethod
    def for_containers(cls, slots, device=None):
        """A decoder for n rANS-W containers that came from somewhere else: shape and stream length are READ from the first
        container's 48-byte header (sicn_codec_parse_header), so a default-constructed LatentCoder — whose stream length is the
        per-image automatic one — is never pointed at containers of another length (the decode would report status bit 2).
        slots: CUDA uint8 [n][slot_bytes].  One small device-to-host copy (the header), once."""
        info = parse_header(bytes(slots[0, :48].cpu().numpy().tobytes()))
        if int(info.mode) != RANSW:
            raise ValueError(f"container mode {int(info.mode)} is not rANS-W")
        coder = cls(int(slots.shape[0]), int(info.lat_h), int(info.lat_w), int(info.lat_c), int(info.image_width), int(info.image_height),
                    device=device if device is not None else slots.device, stream_symbols=int(info.stream_symbols))
        if int(slots.shape[1]) < coder.slot:   # the writer sized its slots for its own stream length; ours can only be equal or smaller
            raise ValueError(f"slots of {int(slots.shape[1])} bytes are shorter than the {coder.slot} bytes a container of this shape may take")
        if int(slots.shape[1]) != coder.slot:   # this object's own container buffer follows the slot size it is told to use: an encode()
            import torch                         # through it must never write past a buffer sized for the smaller slot
            coder.slot = int(slots.shape[1])
            coder.slots = torch.empty((coder.shape[0], coder.slot), dtype=torch.uint8, device=coder.slots.device)
        return coder

    def encode(self, latents, stream=None):
        """latents: CUDA uint8 [n][h][w][c] -> self.slots (containers), self.enc_status. Enqueue only."""
        import torch
        if not (latents.is_cuda and latents.dtype == torch.uint8 and latents.is_contiguous() and tuple(latents.shape) == self.shape):
            raise TypeError(f"latents must be a contiguous CUDA uint8 tensor of shape {self.shape}")
        n, h, w, c = self.shape
        _lib.check(_lib.lib().sicn_codec_encode_batch_async_sl(
            ctypes.c_void_p(latents.data_ptr()), n, w, h, c, self.image_wh[0], self.image_wh[1],
            ctypes.c_void_p(self.slots.data_ptr()), self.slot, ctypes.c_void_p(self.enc_status.data_ptr()),
            ctypes.c_void_p(self.ws.data_ptr()), self.ws.numel(), _stream_ptr(stream), self.stream_symbols), "sicn_codec_encode_batch_async_sl")
        return self.slots

    def decode(self, out_latents, slots=None, valid=None, stream=None):
        """slots (default: self.slots) -> out_latents [n][h][w][c], self.dec_status. `valid`: device int32 [n][2] status
        array whose `.bytes` bound each slot.  Default: this object's own encoder status when it decodes its OWN slots; with
        external `slots` the whole slot is trusted (the header's payload field is validated anyway) unless the caller passes
        the status array that travelled with them.  Pass False to trust the whole slot in every case. Enqueue only."""
        import torch
        n, h, w, c = self.shape
        slots = self.slots if slots is None else slots
        if not (out_latents.is_cuda and out_latents.dtype == torch.uint8 and out_latents.is_contiguous() and tuple(out_latents.shape) == self.shape):
            raise TypeError(f"out_latents must be a contiguous CUDA uint8 tensor of shape {self.shape}")
        if not (slots.is_cuda and slots.dtype == torch.uint8 and slots.is_contiguous() and tuple(slots.shape) == (n, self.slot)):
            raise TypeError("slots must be a contiguous CUDA uint8 tensor [n][slot_bytes]")
        own = slots is self.slots
        vptr = None if (valid is False or (valid is None and not own)) else ctypes.c_void_p((self.enc_status if valid is None else valid).data_ptr())
        _lib.check(_lib.lib().sicn_codec_decode_batch_async_sl(
            ctypes.c_void_p(slots.data_ptr()), self.slot, vptr, n, w, h, c, ctypes.c_void_p(out_latents.data_ptr()), h * w * c,
            ctypes.c_void_p(self.dec_status.data_ptr()), ctypes.c_void_p(self.ws.data_ptr()), self.ws.numel(), _stream_ptr(stream),
            self.stream_symbols), "sicn_codec_decode_batch_async_sl")
        return out_latents

    def sizes(self):
        """Container sizes of the last encode (synchronises)."""
        return [int(v) for v in self.enc_status[:, 1].cpu().tolist()]

    def check(self):
        """Raises SicnError if the last encode / decode reported an error (synchronises)."""
        e = self.enc_status[:, 0].cpu().tolist()
        d = self.dec_status[:, 0].cpu().tolist()
        if any(e):
            raise _lib.SicnError(-22, f"rANS-W encode status {e}")
        if any(d):
            raise _lib.SicnError(-22 if any(v & ~128 for v in d) else -74, f"rANS-W decode status {d}")


class ContextCoder:
    """Container mode 4 ("rANS-WC", sicn_codec_ctx_*_async): the conditional coder of the hyperprior configuration — 16
    class tables, class from the hyper-synthesis scale map and, for the non-anchor half of a checkerboard, from the
    already decoded anchor neighbours.  Same conventions as LatentCoder (enqueue only; verdicts / sizes on the device)."""

    def __init__(self, n_images: int, lat_h: int, lat_w: int, lat_c: int, image_width: int = 0, image_height: int = 0,
                 device="cuda"):
        import torch
        L = _lib.lib()
        self.shape = (int(n_images), int(lat_h), int(lat_w), int(lat_c))
        self.image_wh = (int(image_width), int(image_height))
        self.slot = (int(L.sicn_codec_ctx_max_bytes(lat_w, lat_h, lat_c)) + 255) // 256 * 256
        dev = torch.device(device)
        self.slots = torch.empty((n_images, self.slot), dtype=torch.uint8, device=dev)
        self.ws = torch.empty(max(L.sicn_codec_ctx_workspace_bytes(lat_w, lat_h, lat_c, n_images), 256), dtype=torch.uint8, device=dev)
        self.enc_status = torch.zeros((n_images, 2), dtype=torch.int32, device=dev)
        self.dec_status = torch.zeros((n_images, 2), dtype=torch.int32, device=dev)

    def _check(self, t, what):
        import torch
        if not (t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and tuple(t.shape) == self.shape):
            raise TypeError(f"{what} must be a contiguous CUDA uint8 tensor of shape {self.shape}")

    def encode(self, latents, scales, stream=None):
        self._check(latents, "latents")
        self._check(scales, "scales")
        n, h, w, c = self.shape
        _lib.check(_lib.lib().sicn_codec_ctx_encode_batch_async(
            ctypes.c_void_p(latents.data_ptr()), ctypes.c_void_p(scales.data_ptr()), n, w, h, c, self.image_wh[0], self.image_wh[1],
            ctypes.c_void_p(self.slots.data_ptr()), self.slot, ctypes.c_void_p(self.enc_status.data_ptr()),
            ctypes.c_void_p(self.ws.data_ptr()), self.ws.numel(), _stream_ptr(stream)), "sicn_codec_ctx_encode_batch_async")
        return self.slots

    def decode(self, out_latents, scales, slots=None, valid=None, stream=None):
        self._check(out_latents, "out_latents")
        self._check(scales, "scales")
        n, h, w, c = self.shape
        slots = self.slots if slots is None else slots
        own = slots is self.slots     # external containers: trust the slot unless their status array came along (see LatentCoder.decode)
        vptr = None if (valid is False or (valid is None and not own)) else ctypes.c_void_p((self.enc_status if valid is None else valid).data_ptr())
        _lib.check(_lib.lib().sicn_codec_ctx_decode_batch_async(
            ctypes.c_void_p(slots.data_ptr()), self.slot, vptr, ctypes.c_void_p(scales.data_ptr()), n, w, h, c,
            ctypes.c_void_p(out_latents.data_ptr()), ctypes.c_void_p(self.dec_status.data_ptr()),
            ctypes.c_void_p(self.ws.data_ptr()), self.ws.numel(), _stream_ptr(stream)), "sicn_codec_ctx_decode_batch_async")
        return out_latents

    sizes = LatentCoder.sizes
    check = LatentCoder.check
