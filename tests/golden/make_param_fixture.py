#!/usr/bin/env python3
"""Re-encode the reference's PARAM:: weight/bias tables as a compact numeric fixture.

Reads /root/reference/memdata_nonsquare.h AS TEXT (it is a data table: brace-initialised
`FixedPointWeights<SIMD,ap_int<W>,PE,TILES>` instances, memdata_nonsquare.h:4..207), parses the
integer literals and writes `param_weights.npz` next to this script:

    w{n}_words : uint64 [PE][TILES]   the m_weights words exactly as initialised (weights.hpp:113)
    w{n}_meta  : int32  [4]           (SIMD, W_BIT, PE, TILES)
    b{n}       : int8   [OFM_CH]      bias words reinterpreted as ap_int<8> (conv_nonsquare_top.cpp:272)

Only run in the authoring container (the GPU box has no /root/reference); the .npz is what travels.
"""
import re
import sys
from pathlib import Path

import numpy as np

SRC = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/memdata_nonsquare.h")
OUT = Path(__file__).resolve().parent / "param_weights.npz"

text = SRC.read_text()
decl = re.compile(
    r"static\s+FixedPointWeights<\s*(\d+)\s*,\s*ap_int<\s*(\d+)\s*>\s*,\s*(\d+)\s*,\s*(\d+)\s*>\s*"
    r"(weights|bias)_layer(\d+)\s*=\s*\{")
arrays = {}
matches = list(decl.finditer(text))
for i, m in enumerate(matches):
    simd, wbit, pe, tiles = (int(m.group(k)) for k in range(1, 5))
    kind, layer = m.group(5), int(m.group(6))
    end = matches[i + 1].start() if i + 1 < len(matches) else len(text)
    body = text[m.end():end]
    vals = np.array([int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", body)], dtype=np.uint64)
    assert vals.size == pe * tiles, (kind, layer, vals.size, pe, tiles)
    words = vals.reshape(pe, tiles)
    if kind == "weights":
        assert int(words.max()) < (1 << (simd * wbit))
        arrays[f"w{layer}_words"] = words
        arrays[f"w{layer}_meta"] = np.array([simd, wbit, pe, tiles], dtype=np.int32)
    else:
        assert simd == 1 and wbit == 8 and pe == 1
        arrays[f"b{layer}"] = words.reshape(tiles).astype(np.uint8).view(np.int8)
assert len(arrays) == 8 * 3, sorted(arrays)
np.savez_compressed(OUT, **arrays)
print("wrote", OUT, OUT.stat().st_size, "bytes")
for n in range(8):
    print(n, arrays[f"w{n}_meta"].tolist(), arrays[f"b{n}"].shape,
          "unique weight rows:", len({r.tobytes() for r in arrays[f"w{n}_words"]}))
