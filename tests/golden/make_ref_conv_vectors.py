#!/usr/bin/env python3
"""Golden vectors produced by REFERENCE CODE compiled in the build container.

Runs oracle/_ref/libsicn_refconv.so — oracle/ref_harness.cpp around the reference's own golden
convolution `conv_nonsquare<>` (/root/reference/conv.hpp:91-123, included unmodified; built by
`make -C oracle ref`) — and writes

  tests/golden/ref_conv_vectors.npz   seeded random layers (random nibble weights, biases, pixels >= 128,
                                      odd sizes, Cin/Cout in {3,128,192}): for case k: d{k} (LayerDesc
                                      fields), words{k} (FixedPointWeights tiles), bias{k}, x{k}, y{k}
  tests/golden/ref_conv_hashes.json   SHA-256 of all 8 layer outputs of eight_layers_net (PARAM weights)
                                      for rng256 / ones768 / rng768, each layer computed by conv_nonsquare
                                      exactly as the reference testbench verifies it

This script is the only user of oracle/_ref; it needs /root/reference and therefore runs only in the
build container.  The two files it writes are data (inputs and expected outputs) and travel to the GPU box.

What is reference code here and what is restated:
  * the 7-deep convolution loop, its index order and its 8-bit wrapping accumulation: reference
    (conv.hpp:91-123), TI/TO/TW = uint8_t/int8_t/int8_t for ap_uint<8>/ap_int<8>/ap_int<4>;
  * `output += BIAS; if (output < 0) output = 0`: restated in ref_harness.cpp from conv3_nonsquare_tb.cpp:616-627;
  * the padded input map (tb:581-600) and the zero-stuffed deconv map (tb:700-718): restated below;
  * the FixedPointWeights tile -> W[o][kx][ky][c] walk (tb:546-571): restated below (`tb_unpack_weights`),
    a literal transcription of the testbench's counter walk, NOT the oracle's vectorised unpack.
The dataflow half of the oracle (sliding-window FSM, MVAU fold order) has no reference-compiled anchor:
conv_nonsquare_top.cpp needs the Vivado-HLS headers.
"""
import hashlib
import json
import sys
import time
from dataclasses import astuple
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from simple_image_compression_network_amd.config import LayerDesc, NET_CHANNELS  # noqa: E402

from oracle.ref_conv import dims as _dims, padded_map, ref_layer, tb_unpack_weights  # noqa: E402,F401  (the restated maps / walks live there)


def pack_words(W_okkc, simd, pe):
    """Inverse walk of tb_unpack_weights for W[o][ky][kx][c] nibbles -> words [PE][TILES] (uint64)."""
    cout, _, _, cin = W_okkc.shape
    sf, nf = 25 * cin // simd, cout // pe
    flat = W_okkc.reshape(cout, 25 * cin).astype(np.int64) & 15          # k = (ky*5+kx)*Cin + c
    words = np.zeros((pe, nf * sf), np.uint64)
    for p in range(pe):
        for n in range(nf):
            row = flat[n * pe + p].reshape(sf, simd)
            words[p, n * sf:(n + 1) * sf] = (row.astype(np.uint64) << (np.arange(simd, dtype=np.uint64) * np.uint64(4))).sum(1)
    return words


SMALL = [  # case id, Cin, Cout, SIMD, PE, W, H, transposed
    (0, 3, 128, 3, 8, 24, 16, 0), (1, 3, 128, 3, 8, 21, 17, 0), (2, 128, 128, 8, 16, 20, 16, 0),
    (3, 128, 192, 8, 24, 19, 17, 0), (4, 192, 128, 12, 16, 6, 4, 1), (5, 128, 128, 8, 16, 5, 5, 1),
    (6, 128, 3, 8, 3, 9, 6, 1),
    # folds the net does NOT use (the compiled reference template depends on the dimensions only, so the case ids of rows 2, 5
    # and 3 serve): the tile walk (nf, sf, simd nibble order) is exercised against reference-made bytes at other SIMD / PE too
    (2, 128, 128, 4, 32, 20, 16, 0), (5, 128, 128, 16, 8, 5, 5, 1), (3, 128, 192, 2, 96, 19, 17, 0)]


def main():
    out = {}
    for k, (cid, cin, cout, simd, pe, w, h, tr) in enumerate(SMALL):
        rng = np.random.default_rng(7000 + k)
        ow, oh = (2 * w, 2 * h) if tr else ((w + 1) // 2, (h + 1) // 2)
        d = LayerDesc(IFM_CH=cin, IFM_ROW=w, IFM_COL=h, OFM_CH=cout, OFM_ROW=ow, OFM_COL=oh, SIMD=simd, PE=pe,
                      W_TILES=(cout // pe) * (25 * cin // simd), transposed=tr)
        d.validate()
        W = rng.integers(-8, 8, (cout, 5, 5, cin)).astype(np.int8)
        b = rng.integers(-128, 128, cout).astype(np.int8)
        x = rng.integers(0, 256, (h, w, cin), dtype=np.uint8)
        words = pack_words(W, simd, pe)
        y = ref_layer(cid, x, words, b, simd, pe, tr)
        out[f"d{k}"] = np.array(astuple(d), dtype=np.int32)
        out[f"words{k}"] = words.astype(np.uint32) if simd <= 8 else words
        out[f"bias{k}"] = b
        out[f"x{k}"] = x
        out[f"y{k}"] = y
        print(f"case {k}: {'deconv' if tr else 'conv'} {w}x{h} {cin}->{cout} sha {hashlib.sha256(y.tobytes()).hexdigest()[:16]}")
    np.savez_compressed(HERE / "ref_conv_vectors.npz", **out)

    # whole net, PARAM weights, every layer through conv_nonsquare as conv3_nonsquare_tb.cpp:861-1056 does
    z = np.load(HERE / "param_weights.npz")
    inputs = {"rng256": (10, np.random.default_rng(0).integers(0, 256, (256, 256, 3), dtype=np.uint8)),
              "ones768": (20, np.ones((512, 768, 3), np.uint8)),
              "rng768": (20, np.random.default_rng(0).integers(0, 256, (512, 768, 3), dtype=np.uint8))}
    only = sys.argv[1:]  # optionally restrict to some inputs
    hashes = {"provenance": "conv_nonsquare<> of /root/reference/conv.hpp:91-123 compiled unmodified "
                            "(oracle/ref_harness.cpp, g++ -O2), PARAM weights, layer by layer as "
                            "conv3_nonsquare_tb.cpp:861-1056; made by tests/golden/make_ref_conv_vectors.py",
              "inputs": {}, "layers": {}}
    prev = HERE / "ref_conv_hashes.json"
    if prev.exists():
        old = json.loads(prev.read_text())
        hashes["inputs"].update(old.get("inputs", {}))
        hashes["layers"].update(old.get("layers", {}))
    for name, (base, x) in inputs.items():
        if only and name not in only:
            continue
        hashes["inputs"][name] = hashlib.sha256(x.tobytes()).hexdigest()
        hs = []
        for n, (cin, cout, simd, pe, tr) in enumerate(NET_CHANNELS):
            t0 = time.time()
            x = ref_layer(base + n, x, z[f"w{n}_words"], z[f"b{n}"], simd, pe, tr)
            hs.append(hashlib.sha256(x.tobytes()).hexdigest())
            print(f"{name} L{n}: {x.shape} {time.time() - t0:.1f}s {hs[-1][:16]}", flush=True)
        hashes["layers"][name] = hs
    prev.write_text(json.dumps(hashes, indent=1) + "\n")
    app = json.loads((HERE / "appendix_a_hashes.json").read_text())
    for name, hs in hashes["layers"].items():
        print(name, "== SURVEY Appendix A:", hs == app["layers"][name])


if __name__ == "__main__":
    main()
