"""ORACLE — test infrastructure, NOT product code.

The hyperprior configuration (BASELINE.json configs[4]) restated stage by stage with the C oracle's fast forms, so that it can be
run at 3840 x 2160: layers by oracle/sicn_oracle.c (sicn_or_layer_direct / _direct_act, OpenMP), the activation by
oracle/sicn_gdn_oracle.c, the two containers by oracle/sicn_codec_oracle.c / sicn_hyper_oracle.c.  PARITY UNPINNED (the reference
has none of it).  Used by tests/test_hyperprior.py (4K pipeline test) and tests/golden/make_hyper_hashes.py
(the hashes bench.py's hyperprior leg is held to)."""
from __future__ import annotations

import os

import numpy as np

from . import c_oracle, sicn_ref


def _layer(d, words, bias, x, gdn, threads):
    if gdn is None:
        return c_oracle.run_layer(d, words, bias, x, "direct", threads=threads)
    beta, gamma, inverse, shift = gdn
    return c_oracle.gdn(c_oracle.run_layer_preact(d, words, bias, x, threads=threads), beta, gamma, inverse, shift)


def hyper_pipeline_ref(x: np.ndarray, main_descs, main_words, main_bias, hp: dict, image_wh, z_stream_symbols: int, threads: int | None = None):
    """x: [H][W][3] uint8.  main_*: the eight layers' descs / FINN words / biases; hp: hyperprior.hyper_parameters(...).
    Returns {"y", "z", "z_container", "s", "y_container", "recon"} exactly as HyperpriorCodec produces them for this image."""
    threads = threads or os.cpu_count() or 1
    a = x
    for l in range(4):
        a = _layer(main_descs[l], main_words[l], main_bias[l], a, hp["gdn_np"][l], threads)
    y = a
    z = y
    for d, (w, b) in zip(hp["da"], hp["ha_np"]):
        z = c_oracle.run_layer(d, sicn_ref.pack_finn_tiles(w, d.SIMD, d.PE), b, z, "direct", threads=threads)
    zblob = c_oracle.codec_encode(z, image_wh, 3, stream_symbols=z_stream_symbols)
    z_hat, _ = c_oracle.codec_decode(zblob)
    s = z_hat
    for d, (w, b) in zip(hp["ds"], hp["hs_np"]):
        s = c_oracle.run_layer(d, sicn_ref.pack_finn_tiles(w, d.SIMD, d.PE), b, s, "direct", threads=threads)
    s = np.ascontiguousarray(s[: y.shape[0], : y.shape[1]])
    yblob = c_oracle.ctx_encode(y, s, image_wh)
    y_hat, _ = c_oracle.ctx_decode(yblob, s)
    r = y_hat
    for l in range(4, 8):
        r = _layer(main_descs[l], main_words[l], main_bias[l], r, hp["gdn_np"][l], threads)
    return {"y": y, "z": z, "z_container": zblob, "s": s, "y_container": yblob, "recon": r}
