"""The hyperprior configuration (BASELINE.json configs[4]) assembled from this package's pieces.

EXTENSION BEYOND THE REFERENCE, parity UNPINNED: the reference has no GDN, no hyperprior, no context model and no coder
(SURVEY.md §0).  Everything here is built from operators that DO mirror the reference — `conv2d<>` / `deconv522<>` layers
chained by `sicn_net` — plus this project's own fixed-point GDN (include/sicn_gdn.h) and coders (include/sicn_codec.h):

    main analysis   g_a : x  -> y    L0..L3 of eight_layers_net, GDN in place of the ReLU after L0, L1, L2
    hyper analysis  h_a : y  -> z    conv2d 192->128, conv2d 128->128          (5x5, stride 2, the reference layer template)
    z bitstream               : container mode 3 (rANS-W, one static table measured on z)
    hyper synthesis h_s : z  -> s    deconv522 128->128, deconv522 128->192, cropped to y's shape: the SCALE MAP
    y bitstream               : container mode 4 (rANS-WC: 16 class tables, class from s and — for the non-anchor half of a
                                checkerboard — from the already decoded anchor neighbours)
    main synthesis  g_s : y  -> x^   L4..L7, IGDN after L4, L5, L6

The decoder derives s from the decoded z, the encoder from z itself: the z coder is lossless over bytes, so the two are the same tensor
whenever the container fits its slot (and when it does not, the encoder's status says so and nothing downstream is valid anyway).
`HyperpriorCodec(verify_z=True)` makes the encoder decode its own z container first and use THAT (rounds 2 - 4 always did; it is a
self-check of the coder inside every encode, 0.05 ms of an 8 x 4K step, not something the format needs).
All of it is kernel launches on one stream: no host synchronisation between the stages (verdicts stay on the device until
`check()`), so a whole encode + decode can be enqueued back to back or captured in a hipGraph.

Weights: the main transform uses the PARAM tables (memdata_nonsquare.h); the hyper stacks and the GDN parameters have no
reference values and are seeded random (the reference's own weights are placeholders too)."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import api, codec
from .config import LayerDesc

__all__ = ["HyperpriorCodec", "random_gdn_params", "random_layer_params", "hyper_descs", "hyper_parameters"]


def random_gdn_params(rng, channels: int):
    """beta ~ 1.0 in Q8, sparse small gamma: keeps GDN outputs spread over the int8 range."""
    beta = rng.integers(128, 1024, channels).astype(np.uint32)
    gamma = rng.integers(0, 8, (channels, channels)).astype(np.uint8)
    gamma[rng.random((channels, channels)) < 0.7] = 0
    return beta, gamma


def _pack(w_okkc: np.ndarray, simd: int, pe: int) -> np.ndarray:
    """W[o][ky][kx][c] nibbles -> FixedPointWeights words [PE][TILES] (weights.hpp:110-150; o = nf*PE + pe, k = sf*SIMD + s)."""
    cout, _, _, cin = w_okkc.shape
    kk = 25 * cin
    sf_n, nf_n = kk // simd, cout // pe
    nib = (w_okkc.reshape(nf_n, pe, sf_n, simd).astype(np.int16) & 15).astype(np.uint64)
    shifts = np.arange(simd, dtype=np.uint64) * np.uint64(4)
    words = (nib << shifts[None, None, None, :]).sum(axis=3, dtype=np.uint64)
    return np.ascontiguousarray(words.transpose(1, 0, 2).reshape(pe, nf_n * sf_n))


def random_layer_params(rng, d: LayerDesc):
    """(weights, bias) as api.FixedPointWeights, seeded random nibbles / bytes; also returns the numpy (W, b) for checkers."""
    w = rng.integers(-8, 8, (d.OFM_CH, 5, 5, d.IFM_CH)).astype(np.int8)
    b = rng.integers(-128, 128, d.OFM_CH).astype(np.int8)
    fw = api.FixedPointWeights(d.SIMD, 4, d.PE, d.W_TILES, _pack(w, d.SIMD, d.PE))
    fb = api.FixedPointWeights(1, 8, 1, d.OFM_CH, b.view(np.uint8).astype(np.uint64))
    return (fw, fb), (w, b)


def hyper_descs(lat_w: int, lat_h: int):
    """([h_a descs], [h_s descs]) for a latent of lat_h x lat_w x 192."""
    def mk(cin, cout, simd, pe, w, h, tr):
        ow, oh = (2 * w, 2 * h) if tr else ((w + 1) // 2, (h + 1) // 2)
        d = LayerDesc(IFM_CH=cin, IFM_ROW=w, IFM_COL=h, OFM_CH=cout, OFM_ROW=ow, OFM_COL=oh, SIMD=simd, PE=pe,
                      W_TILES=(cout // pe) * (25 * cin // simd), transposed=tr)
        d.validate()
        return d
    a0 = mk(192, 128, 12, 16, lat_w, lat_h, 0)
    a1 = mk(128, 128, 8, 16, a0.OFM_ROW, a0.OFM_COL, 0)
    s0 = mk(128, 128, 8, 16, a1.OFM_ROW, a1.OFM_COL, 1)
    s1 = mk(128, 192, 8, 24, s0.OFM_ROW, s0.OFM_COL, 1)
    return [a0, a1], [s0, s1]


def hyper_parameters(width: int, height: int, seed: int = 0, use_gdn: bool = True):
    """Every seeded parameter of the configuration, host side only (numpy; no GPU): what HyperpriorCodec uploads and what a checker
    needs to restate the pipeline (tests/golden/make_hyper_hashes.py, tests/test_hyperprior.py).  Returns a dict:
    gdn_np [8] of None | (beta, gamma, inverse, shift); da, ds: the hyper stacks' descs; pa, ps: their (weights, bias) tables;
    ha_np, hs_np: the same as (W[o][ky][kx][c], b).  ONE generator, drawn in this order — the order is part of the seed's meaning."""
    rng = np.random.default_rng(seed)
    descs = api.eight_layer_descs(width, height)
    gdn_np = [None] * 8
    if use_gdn:
        for l in (0, 1, 2, 4, 5, 6):
            beta, gamma = random_gdn_params(rng, descs[l].OFM_CH)
            gdn_np[l] = (beta, gamma, l >= 4, 12)
    lat_h, lat_w, _ = descs[3].out_shape
    da, ds = hyper_descs(lat_w, lat_h)
    pa, ha_np = zip(*[random_layer_params(rng, d) for d in da])
    ps, hs_np = zip(*[random_layer_params(rng, d) for d in ds])
    return {"gdn_np": gdn_np, "da": da, "ds": ds, "pa": list(pa), "ps": list(ps), "ha_np": list(ha_np), "hs_np": list(hs_np)}


class HyperpriorCodec:
    def __init__(self, width: int, height: int, n_images: int, seed: int = 0, device="cuda", use_gdn: bool = True,
                 main_params=None, options=None, z_stream_symbols=None, verify_z: bool = False):
        import torch
        self.n, self.width, self.height = int(n_images), int(width), int(height)
        self.verify_z = bool(verify_z)
        self.device = torch.device(device)
        hp = hyper_parameters(width, height, seed, use_gdn)
        self.gdn_np = hp["gdn_np"]
        gdn = [None if g is None else api.GDN(g[0], g[1], inverse=g[2], shift=g[3]) for g in self.gdn_np]
        self.main = api.EightLayersNet(width, height, params=main_params, device=self.device, gdn=gdn if use_gdn else None,
                                       options=options)
        lat_h, lat_w, lat_c = self.main.descs[3].out_shape
        da, ds, pa, ps = hp["da"], hp["ds"], hp["pa"], hp["ps"]
        self.ha_np, self.hs_np = hp["ha_np"], hp["hs_np"]
        self.h_a = api.EightLayersNet(descs=da, params=list(pa), device=self.device, options=options)
        self.h_s = api.EightLayersNet(descs=ds, params=list(ps), device=self.device, options=options)
        zh, zw, zc = da[-1].out_shape
        n = self.n
        u8 = dict(dtype=torch.uint8, device=self.device)
        self.y = torch.empty((n, lat_h, lat_w, lat_c), **u8)
        self.z = torch.empty((n, zh, zw, zc), **u8)
        self.z_hat = torch.empty_like(self.z)
        self.s_full = torch.empty((n,) + ds[-1].out_shape, **u8)     # 2 * ceil(./2) >= the latent's size: cropped below
        self.s = torch.empty_like(self.y)
        self.y_hat = torch.empty_like(self.y)
        # the hyper-latent is small (4K: 261 k symbols): codec.auto_stream_symbols picks its stream length from ONE image's latent,
        # so a container set written by a batch of 8 decodes in a batch of 1; `z_stream_symbols` overrides it and is
        # then part of the format the decoder must be built with
        self.z_coder = codec.LatentCoder(n, zh, zw, zc, width, height, device=self.device, stream_symbols=z_stream_symbols or "auto")
        self.y_coder = codec.ContextCoder(n, lat_h, lat_w, lat_c, width, height, device=self.device)
        for net in (self.main, self.h_a, self.h_s):
            net.workspace(n)

    def _scale_map(self, z):
        """h_s(z) cropped to the latent's shape (deconv doubles a size that conv rounded up)."""
        import ctypes
        self.h_s.run_layers(0, 1, z, out=self.s_full)
        n, h, w, c = self.y.shape
        _, hs, ws, _ = self.s_full.shape
        # one launch for the batch (torch's strided copy_ was one D2D memcpy per image: ~0.8 ms of an 8 x 4K encode + decode step)
        api._lib.check(api._lib.lib().sicn_crop_nhwc(ctypes.c_void_p(self.s_full.data_ptr()), ctypes.c_void_p(self.s.data_ptr()), n, hs,
                                                     ws, h, w, c, api._stream_ptr(None)), "sicn_crop_nhwc")
        return self.s

    def encode(self, x):
        """x: CUDA uint8 [n][H][W][3] -> two sets of containers (self.z_coder.slots, self.y_coder.slots). Enqueue only."""
        self.main.analysis(x, self.y)
        self.h_a.run_layers(0, 1, self.y, out=self.z)
        self.z_coder.encode(self.z)
        # the model must be the one the decoder can rebuild: h_s of the hyper-latent the container holds — which IS z (lossless coder);
        # verify_z decodes the container to prove it on every call
        if self.verify_z:
            self.z_coder.decode(self.z_hat)
        self.y_coder.encode(self.y, self._scale_map(self.z_hat if self.verify_z else self.z))
        return self.z_coder.slots, self.y_coder.slots

    def decode(self, out, z_slots=None, y_slots=None, z_valid=None, y_valid=None):
        """containers -> reconstruction `out` [n][H'][W'][3]. Enqueue only.  z_valid / y_valid: the encoder's status arrays
        (device int32 [n][2], `.bytes` bounds each container) when they travelled with external slots; without them an
        external slot is trusted as a whole (the container header is validated either way)."""
        self.z_coder.decode(self.z_hat, slots=z_slots, valid=z_valid)
        self.y_coder.decode(self.y_hat, self._scale_map(self.z_hat), slots=y_slots, valid=y_valid)
        self.main.synthesis(self.y_hat, out)
        return out

    def check(self):
        self.z_coder.check()
        self.y_coder.check()

    def bytes_per_image(self):
        return [a + b for a, b in zip(self.z_coder.sizes(), self.y_coder.sizes())]
