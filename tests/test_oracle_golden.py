"""Pins the oracle (oracle/sicn_ref.py + oracle/sicn_oracle.c) before anything trusts it.

Known answers: SURVEY.md Appendix A — SHA-256 of all 8 layer outputs, for three inputs, produced
by the reference's own conv2d<>/deconv522<> templates with the PARAM:: weights
(tests/golden/appendix_a_hashes.json).  Plus the reference's own self-check restated
(dataflow == naive golden on the all-ones stimulus, conv3_nonsquare_tb.cpp:1068-1104).
"""
import hashlib
import json

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import c_oracle, sicn_ref
from simple_image_compression_network_amd.config import LayerDesc, REFERENCE_DESCS, eight_layer_descs

HASHES = json.loads((GOLDEN / "appendix_a_hashes.json").read_text())


def _input(name):
    if name == "ones768":
        x = np.ones((512, 768, 3), np.uint8)         # conv3_nonsquare_tb.cpp:801
    elif name == "rng768":
        x = np.random.default_rng(0).integers(0, 256, (512, 768, 3), dtype=np.uint8)
    else:
        x = np.random.default_rng(0).integers(0, 256, (256, 256, 3), dtype=np.uint8)
    assert hashlib.sha256(x.tobytes()).hexdigest() == HASHES["inputs"][name]["sha256"]
    return x


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", ["ones768", "rng768", "rng256"])
def test_closed_form_matches_reference_hashes(name, param_closed_form):
    outs = sicn_ref.eight_layers_net_ref(_input(name), param_closed_form)
    assert [_sha(o) for o in outs] == HASHES["layers"][name]


def test_reference_config_shapes():
    # config_nonsquare.h:1-135
    assert [(d.IFM_ROW, d.IFM_COL, d.IFM_CH) for d in REFERENCE_DESCS] == [
        (768, 512, 3), (384, 256, 128), (192, 128, 128), (96, 64, 128),
        (48, 32, 192), (96, 64, 128), (192, 128, 128), (384, 256, 128)]
    assert [d.W_TILES for d in REFERENCE_DESCS] == [400, 3200, 3200, 3200, 3200, 3200, 3200, 400]
    assert [(d.SIMD, d.PE) for d in REFERENCE_DESCS] == [
        (3, 8), (8, 16), (8, 16), (8, 24), (12, 16), (8, 16), (8, 16), (8, 3)]


def test_c_direct_matches_reference_hashes_rng256(param_words):
    descs = eight_layer_descs(256, 256)
    outs = c_oracle.run_net(descs, [w for w, _ in param_words], [b for _, b in param_words],
                            _input("rng256"), "direct", threads=8)
    assert [_sha(o) for o in outs] == HASHES["layers"]["rng256"]


def test_c_dataflow_matches_reference_hashes_analysis_rng256(param_words):
    """The stage-by-stage dataflow port (FSM sliding window, 8-bit wrapping MVAU) on BASELINE
    config 2 (256x256, 4-layer analysis) against the reference's own latent hash."""
    descs = eight_layer_descs(256, 256)[:4]
    outs = c_oracle.run_net(descs, [w for w, _ in param_words][:4], [b for _, b in param_words][:4],
                            _input("rng256"), "dataflow")
    assert [_sha(o) for o in outs] == HASHES["layers"]["rng256"][:4]


@pytest.mark.slow
def test_c_dataflow_matches_reference_hashes_synthesis_rng256(param_words, param_closed_form):
    descs = eight_layer_descs(256, 256)
    latent = sicn_ref.eight_layers_net_ref(_input("rng256"), param_closed_form)[3]
    outs = c_oracle.run_net(descs[4:], [w for w, _ in param_words][4:], [b for _, b in param_words][4:],
                            latent, "dataflow")
    assert [_sha(o) for o in outs] == HASHES["layers"]["rng256"][4:]


def test_reference_selfcheck_restated_all_ones(param_words):
    """conv3_nonsquare_tb.cpp:781-1125 at reduced size: all-ones image through the dataflow port
    vs the testbench's golden model (tb weight unpack + conv_nonsquare + bias/ReLU), every layer."""
    descs = eight_layer_descs(48, 32)
    x = np.ones((32, 48, 3), np.uint8)
    words = [w for w, _ in param_words]
    bias = [b for _, b in param_words]
    df = c_oracle.run_net(descs, words, bias, x, "dataflow")
    x_in = x
    for n, d in enumerate(descs):           # golden chain feeds on its OWN previous output (tb:900-1056)
        g = c_oracle.run_layer(d, words[n], bias[n], x_in, "naive")
        assert np.array_equal(g, df[n]), f"layer {n}"
        x_in = g


def _rand_layer(rng, cin, cout, simd, pe, w, h, tr):
    ow, oh = (2 * w, 2 * h) if tr else ((w + 1) // 2, (h + 1) // 2)
    d = LayerDesc(IFM_CH=cin, IFM_ROW=w, IFM_COL=h, OFM_CH=cout, OFM_ROW=ow, OFM_COL=oh, SIMD=simd,
                  PE=pe, W_TILES=(cout // pe) * (25 * cin // simd), transposed=tr)
    d.validate()
    W = rng.integers(-8, 8, (cout, 5, 5, cin)).astype(np.int8)
    b = rng.integers(-128, 128, cout).astype(np.int8)
    x = rng.integers(0, 256, (h, w, cin), dtype=np.uint8)          # includes pixels >= 128
    return d, W, b, x


CASES = [(3, 8, 3, 4, 20, 12, 0), (6, 6, 2, 3, 13, 9, 0), (4, 12, 4, 6, 7, 5, 1), (12, 3, 12, 3, 9, 11, 1),
         (6, 4, 3, 2, 12, 18, 0), (128, 128, 8, 16, 10, 6, 0), (128, 192, 8, 24, 6, 4, 0),
         (192, 128, 12, 16, 5, 3, 1), (128, 3, 8, 3, 6, 7, 1), (3, 128, 3, 8, 17, 11, 0)]


@pytest.mark.parametrize("case", CASES)
def test_all_forms_agree_random_weights(case):
    """Seeded random nibble weights / biases / u8 pixels incl. >=128, odd sizes, several folds:
    dataflow (FSM), dataflow (im2col), testbench golden, C closed form and numpy closed form are
    byte-identical — a channel-permutation or lane-order slip cannot hide behind the reference's
    all-PEs-equal placeholder weights."""
    rng = np.random.default_rng(hash(case) % (1 << 32))
    d, W, b, x = _rand_layer(rng, *case)
    words = sicn_ref.pack_finn_tiles(W, d.SIMD, d.PE)
    assert np.array_equal(sicn_ref.unpack_finn_tiles(words, d.SIMD, d.PE, d.IFM_CH, d.OFM_CH), W)
    ref = (sicn_ref.deconv522_ref if d.transposed else sicn_ref.conv2d_ref)(x, W, b)
    assert ref.max() <= 127
    for form in ("dataflow", "dataflow_im2col", "naive", "direct"):
        assert np.array_equal(c_oracle.run_layer(d, words, b, x, form), ref), form


def test_sliding_window_fsm_valid_domain():
    """slidingwindow.h:1297,1327,1345-1348 (and the `// TODO 18*12 is error` at
    conv_nonsquare_top.cpp:238): the FSM is a correct stride-1 im2col exactly while
    OFMDim_y <= IFMDim_x * (C/SIMD) (padded dims); it first diverges at +1."""
    L = c_oracle.lib()
    rng = np.random.default_rng(0)

    def trial(px, py, ch, simd):
        sx, sy = px - 4, py - 4
        x = rng.integers(0, 256, (py, px, ch), dtype=np.uint8)
        a = np.zeros(sy * sx * 25 * ch, np.uint8)
        b = np.zeros_like(a)
        n = L.sicn_or_swg_nonsquare_fsm(c_oracle._ptr(x), px * py * (ch // simd), c_oracle._ptr(a),
                                        5, 5, ch, px, py, sx, sy, simd, 1, 1)
        L.sicn_or_im2col_s1(c_oracle._ptr(x), c_oracle._ptr(b), 5, ch, px, sx, sy)
        return n == sy * sx * 25 * (ch // simd) and np.array_equal(a, b)

    assert trial(12, 16, 1, 1) and not trial(12, 17, 1, 1)
    assert trial(12, 28, 2, 1) and not trial(12, 29, 2, 1)
    assert trial(260, 260, 3, 3)                     # BASELINE config 2, layer 0: 256 <= 260


def test_oracle_rejects_bad_descs(param_words):
    d = eight_layer_descs(32, 16)[1]
    bad = LayerDesc(**{**d.__dict__, "SIMD": 7})
    with pytest.raises(RuntimeError):
        c_oracle.run_layer(bad, param_words[1][0], param_words[1][1],
                           np.zeros(d.in_shape, np.uint8), "direct")


def _small_vectors(name="small_vectors.npz"):
    from dataclasses import fields
    z = np.load(GOLDEN / name)
    k = 0
    while f"d{k}" in z:
        d = LayerDesc(**{f.name: int(v) for f, v in zip(fields(LayerDesc), z[f"d{k}"])})
        yield d, z[f"words{k}"], z[f"bias{k}"], z[f"x{k}"], z[f"y{k}"]
        k += 1


def test_committed_small_vectors_still_hold():
    """tests/golden/small_vectors.npz (made by make_small_vectors.py from the pinned oracle): every CPU form
    must still reproduce the stored outputs byte for byte."""
    n = 0
    for d, words, bias, x, y in _small_vectors():
        for form in ("dataflow_im2col", "naive", "direct"):
            assert np.array_equal(c_oracle.run_layer(d, words, bias, x, form), y), (d, form)
        n += 1
    assert n == 7


# ---- fixtures made by REFERENCE code: conv_nonsquare<> of /root/reference/conv.hpp:91-123, compiled unmodified
# ---- (oracle/ref_harness.cpp -> oracle/_ref, tests/golden/make_ref_conv_vectors.py). Only the data travels.
REF_HASHES = json.loads((GOLDEN / "ref_conv_hashes.json").read_text())


def test_reference_compiled_hashes_cover_appendix_a():
    """All 24 known answers of SURVEY Appendix A were re-made here by the reference's own golden convolution
    (layer by layer as conv3_nonsquare_tb.cpp:861-1056 does): the oracle's pin is reference code, not the survey."""
    assert REF_HASHES["inputs"] == {k: v["sha256"] for k, v in HASHES["inputs"].items()}
    for name in ("ones768", "rng768", "rng256"):
        assert REF_HASHES["layers"][name] == HASHES["layers"][name], name


@pytest.mark.parametrize("name", ["ones768", "rng768", "rng256"])
def test_closed_form_matches_reference_compiled_hashes(name, param_closed_form):
    outs = sicn_ref.eight_layers_net_ref(_input(name), param_closed_form)
    assert [_sha(o) for o in outs] == REF_HASHES["layers"][name]


def test_every_oracle_form_matches_reference_compiled_vectors():
    """tests/golden/ref_conv_vectors.npz: seeded random nibble weights / biases / pixels >= 128, odd sizes,
    Cin/Cout in {3,128,192}, outputs computed by the reference's conv_nonsquare<> on the testbench's padded /
    zero-stuffed maps (conv3_nonsquare_tb.cpp:581-600,700-718). Random weights differ per PE, so a channel or
    SIMD-lane permutation in the oracle's tile unpack would show here (the PARAM tables cannot show it)."""
    n = 0
    for d, words, bias, x, y in _small_vectors("ref_conv_vectors.npz"):
        w = sicn_ref.unpack_finn_tiles(words, d.SIMD, d.PE, d.IFM_CH, d.OFM_CH)
        got = (sicn_ref.deconv522_ref if d.transposed else sicn_ref.conv2d_ref)(x, w, bias)
        assert np.array_equal(got, y), (d, "numpy closed form")
        for form in ("dataflow", "dataflow_im2col", "naive", "direct"):
            assert np.array_equal(c_oracle.run_layer(d, words, bias, x, form), y), (d, form)
        n += 1
    assert n == 10


def test_ref_conv_padded_map_forms_agree():
    """oracle/ref_conv.py keeps the testbench's zero-stuffed deconv map twice: as the testbench's loops (tb:700-718) and
    vectorised (what the fixtures script and bench.py's reference leg call)."""
    from oracle import ref_conv
    rng = np.random.default_rng(5)
    for shape in [(1, 1, 3), (7, 5, 3), (4, 9, 16), (6, 6, 1)]:
        x = rng.integers(0, 256, shape, dtype=np.uint8)
        for tr in (0, 1):
            assert np.array_equal(ref_conv.padded_map(x, tr), ref_conv.padded_map_loops(x, tr)), (shape, tr)
