#!/usr/bin/env python3
"""Small committed input/output vectors for every kernel family, made by the PINNED oracle
(oracle/sicn_ref.py, itself pinned to SURVEY.md Appendix A by tests/test_oracle_golden.py) with
seeded random nibble weights / biases / pixels.  They freeze today's answers: a later change to
the oracle OR to the kernels that alters a byte shows up against these files.

writes tests/golden/small_vectors.npz: for case k: d{k} (LayerDesc fields), words{k}, bias{k}, x{k}, y{k}."""
import sys
from dataclasses import astuple
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import sicn_ref  # noqa: E402
from simple_image_compression_network_amd.config import LayerDesc  # noqa: E402

CASES = [(3, 128, 3, 8, 37, 21, 0), (128, 128, 8, 16, 35, 11, 0), (128, 192, 8, 24, 33, 9, 0),
         (192, 128, 12, 16, 9, 5, 1), (128, 128, 8, 16, 34, 7, 1), (128, 3, 8, 3, 33, 6, 1), (6, 4, 3, 2, 12, 9, 0)]
out = {}
for k, (cin, cout, simd, pe, w, h, tr) in enumerate(CASES):
    rng = np.random.default_rng(1000 + k)
    ow, oh = (2 * w, 2 * h) if tr else ((w + 1) // 2, (h + 1) // 2)
    d = LayerDesc(IFM_CH=cin, IFM_ROW=w, IFM_COL=h, OFM_CH=cout, OFM_ROW=ow, OFM_COL=oh, SIMD=simd, PE=pe,
                  W_TILES=(cout // pe) * (25 * cin // simd), transposed=tr)
    d.validate()
    W = rng.integers(-8, 8, (cout, 5, 5, cin)).astype(np.int8)
    b = rng.integers(-128, 128, cout).astype(np.int8)
    x = rng.integers(0, 256, (h, w, cin), dtype=np.uint8)
    y = (sicn_ref.deconv522_ref if tr else sicn_ref.conv2d_ref)(x, W, b)
    out[f"d{k}"] = np.array(astuple(d), dtype=np.int32)
    words = sicn_ref.pack_finn_tiles(W, simd, pe)
    out[f"words{k}"] = words.astype(np.uint32) if simd <= 8 else words   # narrowest dtype that holds SIMD nibbles
    out[f"bias{k}"] = b
    out[f"x{k}"] = x
    out[f"y{k}"] = y
np.savez_compressed(Path(__file__).resolve().parent / "small_vectors.npz", **out)
print("wrote small_vectors.npz,", len(CASES), "cases")
