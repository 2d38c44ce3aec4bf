"""ORACLE PIN — test infrastructure, NOT product code: the Python side of oracle/_ref/libsicn_refconv.so, i.e. of the
reference's own golden convolution `conv_nonsquare<>` (/root/reference/conv.hpp:91-123, compiled UNMODIFIED by
`make -C oracle ref` from oracle/ref_harness.cpp).  Users: tests/golden/make_ref_conv_vectors.py (fixtures, build container)
and bench.py's `cpu_baseline` leg (the reference's CPU model timed on the GPU box's host: the built .so travels with the
repository snapshot, the reference's sources do not and are not needed at run time).

What is reference code here and what is restated: see tests/golden/make_ref_conv_vectors.py's header; in short the 7-deep
loop and its 8-bit wrapping accumulation are the reference's, the padded / zero-stuffed input maps (conv3_nonsquare_tb.cpp:
581-600, 700-718) and the tile -> W[o][kx][ky][c] walk (tb:546-571) are transcriptions of the testbench."""
import ctypes
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
SO = ROOT / "oracle" / "_ref" / "libsicn_refconv.so"
_LIB = None

# case ids instantiated in ref_harness.cpp for whole nets: base + layer
NET_256, NET_768x512 = 10, 20


def available() -> bool:
    return SO.exists()


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(str(SO))
        _LIB.sicn_refconv_dims.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        _LIB.sicn_refconv_run.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4
    return _LIB


def dims(case_id):
    d = (ctypes.c_int * 7)()
    assert lib().sicn_refconv_dims(case_id, d) == 0, case_id
    return tuple(d)


def padded_map_loops(x, transposed):
    """x [H][W][C] -> the testbench's input_padding[x][y][c] (index order x FIRST), written as the testbench writes it.
    conv (tb:581-600): 2-pixel zero border.  deconv (tb:700-718): size 2W+4 x 2H+4, zero where
    ox < 2 | ox >= 2W+2 | (ox-1) even (same for y), else input[(ox-2)/2][(oy-2)/2]."""
    h, w, c = x.shape
    if not transposed:
        p = np.zeros((w + 4, h + 4, c), np.uint8)
        p[2:2 + w, 2:2 + h] = x.transpose(1, 0, 2)
        return p
    p = np.zeros((2 * w + 4, 2 * h + 4, c), np.uint8)
    for ox in range(2 * w + 4):
        if ox < 2 or ox >= 2 * w + 2 or (ox + 1 - 2) % 2 == 0:
            continue
        for oy in range(2 * h + 4):
            if oy < 2 or oy >= 2 * h + 2 or (oy + 1 - 2) % 2 == 0:
                continue
            p[ox, oy] = x[(oy - 2) // 2, (ox - 2) // 2]
    return p


def padded_map(x, transposed):
    """The same map without Python loops (tests/test_oracle_golden.py holds the two forms equal): the deconv map keeps
    input (i, j) at [2 + 2 j][2 + 2 i]."""
    if not transposed:
        return padded_map_loops(x, False)
    h, w, c = x.shape
    p = np.zeros((2 * w + 4, 2 * h + 4, c), np.uint8)
    p[2:2 * w + 2:2, 2:2 * h + 2:2] = x.transpose(1, 0, 2)
    return p


def tb_unpack_weights(words, simd, pe, cin, cout):
    """conv3_nonsquare_tb.cpp:546-571 transcribed: words[PE][TILES] -> W[o][kx][ky][c] (int8).
    `weights(tile)[pe][simd]` = sign-extended nibble `simd` of m_weights[pe][tile] (weights.hpp:134-139)."""
    tx, ty = (cin * 25) // simd, cout // pe
    W = np.zeros((cout, 5, 5, cin), np.int8)
    kx = ky = chan = 0
    for p in range(pe):
        o = p
        for oy in range(ty):
            for ox in range(tx):
                word = int(words[p][oy * tx + ox])
                for s in range(simd):
                    n = (word >> (4 * s)) & 15
                    W[o, kx, ky, chan] = n - 16 if n > 7 else n
                    chan += 1
                    if chan == cin:
                        chan = 0
                        kx += 1
                        if kx == 5:
                            kx = 0
                            ky += 1
                            if ky == 5:
                                ky = 0
                                o += pe
                                if o == cout:
                                    o = 0
    return W


def ref_layer(case_id, x, words, bias, simd, pe, transposed, seconds=None):
    """One layer through the reference's conv_nonsquare. x [H][W][Cin] uint8 -> [OH][OW][Cout] uint8.
    `seconds` (a list) receives the time spent inside the reference code (sicn_refconv_run) only."""
    ix, iy, ox, oy, ci, co, s = dims(case_id)
    h, w, c = x.shape
    assert c == ci and s == (1 if transposed else 2)
    assert (ix, iy) == ((2 * w + 4, 2 * h + 4) if transposed else (w + 4, h + 4)), (case_id, ix, iy, x.shape)
    assert (ox, oy) == ((2 * w, 2 * h) if transposed else ((w + 1) // 2, (h + 1) // 2))
    img = np.ascontiguousarray(padded_map(x, transposed))
    W = np.ascontiguousarray(tb_unpack_weights(words, simd, pe, ci, co))
    b = np.ascontiguousarray(bias, dtype=np.int8)
    out = np.zeros((ox, oy, co), np.int8)
    t0 = time.perf_counter()
    rc = lib().sicn_refconv_run(case_id, img.ctypes.data, W.ctypes.data, b.ctypes.data, out.ctypes.data)
    if seconds is not None:
        seconds.append(time.perf_counter() - t0)
    assert rc == 0
    assert out.min() >= 0
    return np.ascontiguousarray(out.transpose(1, 0, 2)).view(np.uint8)


def run_net(base_case, image, words, bias, net_channels):
    """eight_layers_net layer by layer through conv_nonsquare (as conv3_nonsquare_tb.cpp:861-1056 does) on an image of the
    size the case set `base_case` was instantiated for.  Returns (outputs of all 8 layers, seconds inside the reference code)."""
    secs, outs, x = [], [], image
    for n, (cin, cout, simd, pe, tr) in enumerate(net_channels):
        x = ref_layer(base_case + n, x, words[n], bias[n], simd, pe, tr, secs)
        outs.append(x)
    return outs, sum(secs)
