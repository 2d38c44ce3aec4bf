"""Generic ConvLayer_Batch surface (convlayer.h:89-125).  The reference never runs
it (conv_nonsquare_top.cpp:223 is commented out), so parity is unpinned: the CPU tests show the dataflow
restatement (square sliding-window FSM + folded MVAU with a wrapping TA accumulator + activation functor)
equals the closed form; the GPU tests show the HIP kernel equals both."""
import ctypes
import re
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import c_oracle, sicn_ref

ROOT = Path(__file__).resolve().parent.parent

# (K, C, D, O, SIMD, PE, W_BIT, IN_SIGNED, ACC_BIT, ACC_SIGNED, OUT_BIT, NUM_TH, ACT_VAL)
CASES = [
    (3, 4, 9, 6, 2, 3, 4, 0, 16, 1, 16, 0, 0),       # BNN-pynq style conv, 16-bit signed accumulator passed through
    (5, 3, 12, 8, 3, 4, 4, 0, 8, 0, 8, 0, 0),        # the net's own arithmetic (ap_uint<8> accumulator) without padding
    (1, 8, 5, 4, 8, 2, 8, 1, 32, 1, 32, 0, 0),       # 1x1, signed inputs, 8-bit weights, full 32-bit result
    (3, 6, 8, 4, 3, 2, 2, 1, 12, 1, 16, 0, 0),       # 2-bit weights, 12-bit wrapping accumulator
    (3, 4, 9, 6, 4, 2, 4, 0, 16, 1, 8, 3, 0),        # 3 thresholds -> 2-bit activations
    (2, 8, 7, 12, 8, 6, 3, 1, 10, 1, 8, 15, -8),     # 15 thresholds, ActVal -8 (signed 4-bit result in a byte lane)
    (7, 2, 10, 2, 2, 1, 5, 0, 24, 0, 32, 1, 0),      # one threshold (sign), unsigned accumulator compare
    # IFM_CH % 16 == 0: served by the MFMA kernel on the GPU (the others by the direct one)
    (3, 16, 9, 6, 8, 3, 4, 0, 16, 1, 16, 0, 0),      # one 16-byte channel group, O not a multiple of 16, unsigned inputs
    (1, 64, 6, 40, 8, 8, 8, 1, 32, 1, 32, 0, 0),     # 1x1, signed inputs, 8-bit weights, 3 channel tiles (one partial)
    (3, 48, 10, 20, 8, 4, 4, 0, 8, 0, 8, 0, 0),      # C = 48: a partial 64-byte K step; the net's ap_uint<8> accumulator
    (5, 32, 12, 70, 8, 7, 3, 0, 14, 1, 8, 7, -4),    # 7 thresholds, 70 channels: two grid rows of channel tiles
    (2, 128, 20, 16, 8, 8, 2, 1, 12, 1, 16, 0, 0),   # two full K steps per tap, 361 positions: partial position tile
]


def _make(case, rng):
    K, C, D, O, SIMD, PE, WB, INS, AB, AS, OB, NTH, AV = case
    nf = O // PE
    w = rng.integers(-(1 << (WB - 1)), 1 << (WB - 1), (O, K * K * C)).astype(np.int8)
    words = sicn_ref.pack_finn_tiles_generic(w, SIMD, PE, WB)
    x = rng.integers(0, 256, (D, D, C), dtype=np.uint8)
    thr = None
    if NTH:
        lim = min(1 << (AB - 1), 4000)
        thr = np.sort(rng.integers(-lim if AS else 0, lim, (PE, nf, NTH)), axis=2).astype(np.int32)
    desc = SimpleNamespace(K=K, IFM_CH=C, IFM_DIM=D, OFM_CH=O, OFM_DIM=D - K + 1, SIMD=SIMD, PE=PE, IN_BIT=8, IN_SIGNED=INS,
                           W_BIT=WB, W_TILES=nf * (K * K * C // SIMD), ACC_BIT=AB, ACC_SIGNED=AS, OUT_BIT=OB,
                           activation=int(NTH > 0), NUM_TH=NTH, ACT_VAL=AV)
    thr_oi = None if thr is None else thr.transpose(1, 0, 2).reshape(O, NTH)     # [pe][nf] -> o = nf*PE + pe
    ref = sicn_ref.conv_layer_batch_ref(x, w, K, bool(INS), AB, bool(AS), OB, thr_oi, AV)
    return desc, w, words, x, thr, ref


@pytest.mark.parametrize("case", CASES)
def test_dataflow_restatement_equals_closed_form(case):
    rng = np.random.default_rng(abs(hash(case)) % (1 << 32))
    desc, _, words, x, thr, ref = _make(case, rng)
    for fsm in (True, False):
        assert np.array_equal(c_oracle.convlayer_dataflow(desc, words, thr, x, use_fsm=fsm), ref)


def test_net_layer_without_padding_is_a_special_case(param_words):
    """conv2d<> = pad + ConvLayer-style sliding window + decimation: feeding the PADDED image to the generic
    layer (K=5, ap_uint<8> accumulator, pass-through) and keeping even rows/cols, then bias/ReLU, must give
    layer 0 of the net (conv_nonsquare_top.cpp:198-280)."""
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (20, 20, 3), dtype=np.uint8)
    words, bias = param_words[0]
    w = sicn_ref.unpack_finn_tiles(words, 3, 8, 3, 128)
    xpad = np.pad(x, ((2, 2), (2, 2), (0, 0)))
    full = sicn_ref.conv_layer_batch_ref(xpad, w.reshape(128, 75), 5, False, 8, False, 8)
    v = (full[::2, ::2].astype(np.int64) + bias.astype(np.int64)) & 0xFF
    v[v >= 128] = 0
    assert np.array_equal(v.astype(np.uint8), sicn_ref.conv2d_ref(x, w, bias))


def test_convlayer_abi_symbols_and_validation():
    from simple_image_compression_network_amd import _lib
    text = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "sicn_convlayer.h").read_text(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(sicn_conv[a-z0-9_]*)\s*\(", text)))
    L = _lib.lib()        # (loads the HIP runtime PyTorch ships first, then libsicn.so)
    assert set(syms) == set(_lib.CONVLAYER_ABI) and all(hasattr(L, s) for s in syms)
    from simple_image_compression_network_amd.convlayer import ConvLayerDesc, PassThroughActivation
    good = ConvLayerDesc(K=3, IFM_CH=4, IFM_DIM=9, OFM_CH=6, SIMD=2, PE=3).to_c(PassThroughActivation(16, True))
    assert _lib.lib().sicn_convlayer_validate(ctypes.byref(good)) == 0
    for field, val in (("K", 12), ("SIMD", 3), ("PE", 4), ("W_BIT", 9), ("ACC_BIT", 33), ("OUT_BIT", 12), ("OFM_DIM", 9),
                       ("NUM_TH", 2), ("IN_BIT", 3), ("IN_BIT", 1), ("OUT_BIT", 2)):      # IN_BIT 1 x 4 channels / OUT_BIT 2 x 6: not whole bytes
        bad = ConvLayerDesc(K=3, IFM_CH=4, IFM_DIM=9, OFM_CH=6, SIMD=2, PE=3).to_c(PassThroughActivation(16, True))
        setattr(bad, field, val)
        assert _lib.lib().sicn_convlayer_validate(ctypes.byref(bad)) == -22, field


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_conv_layer_batch_matches_oracle(case):
    import torch
    from simple_image_compression_network_amd.api import FixedPointWeights
    from simple_image_compression_network_amd.convlayer import (ConvLayer_Batch, ConvLayerDesc, PassThroughActivation,
                                                                ThresholdsActivation)
    rng = np.random.default_rng(abs(hash(case)) % (1 << 32))
    K, C, D, O, SIMD, PE, WB, INS, AB, AS, OB, NTH, AV = case
    d, _, words, x, thr, ref = _make(case, rng)
    desc = ConvLayerDesc(K=K, IFM_CH=C, IFM_DIM=D, OFM_CH=O, SIMD=SIMD, PE=PE, W_BIT=WB, IN_SIGNED=bool(INS), OUT_BIT=OB)
    act = ThresholdsActivation(thr, AB, bool(AS), AV) if NTH else PassThroughActivation(AB, bool(AS))
    xin = torch.from_numpy(np.stack([x, x[::-1].copy()])).cuda()
    out = ConvLayer_Batch(desc, xin, None, FixedPointWeights(SIMD, WB, PE, d.W_TILES, words), act, 2)
    got = out.cpu().numpy().astype(np.int64) & ((1 << OB) - 1)
    assert np.array_equal(got[0], ref.astype(np.int64))
    ref1 = c_oracle.convlayer_dataflow(d, words, thr, x[::-1].copy(), use_fsm=False)
    assert np.array_equal(got[1], ref1.astype(np.int64))


@pytest.mark.gpu
def test_gpu_mfma_and_direct_kernels_agree():
    """The same layer through k_convlayer_mfma (default when IFM_CH % 16 == 0) and through the direct kernel
    (sicn_conv_layer_batch_kernel(..., SICN_CONVLAYER_KERNEL_DIRECT)), on a shape large enough for several workgroups."""
    import torch
    from simple_image_compression_network_amd.api import FixedPointWeights
    from simple_image_compression_network_amd.convlayer import ConvLayer_Batch, ConvLayerDesc, PassThroughActivation
    rng = np.random.default_rng(77)
    K, C, D, O = 3, 64, 40, 96
    w = rng.integers(-8, 8, (O, K * K * C)).astype(np.int8)
    words = sicn_ref.pack_finn_tiles_generic(w, 8, 8, 4)
    desc = ConvLayerDesc(K=K, IFM_CH=C, IFM_DIM=D, OFM_CH=O, SIMD=8, PE=8, W_BIT=4, IN_SIGNED=False, OUT_BIT=32)
    act = PassThroughActivation(24, True)
    x = torch.from_numpy(rng.integers(0, 256, (3, D, D, C), dtype=np.uint8)).cuda()
    fpw = FixedPointWeights(8, 4, 8, desc.W_TILES, words)
    a = ConvLayer_Batch(desc, x, None, fpw, act, 3).clone()
    b = ConvLayer_Batch(desc, x, None, fpw, act, 3, kernel=1)
    assert torch.equal(a, b)
    ref = sicn_ref.conv_layer_batch_ref(x[1].cpu().numpy(), w, K, False, 24, True, 32)
    assert np.array_equal(a[1].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, ref.astype(np.int64))


@pytest.mark.gpu
def test_gpu_resident_conv_layer_equals_the_per_call_form():
    """ConvLayer keeps descriptor, weights and thresholds on the device; repeated calls (no synchronisation in between, a second stream)
    give what ConvLayer_Batch — upload, run, free per call: the reference's calling convention — gives."""
    import torch
    from simple_image_compression_network_amd.api import FixedPointWeights
    from simple_image_compression_network_amd.convlayer import ConvLayer, ConvLayer_Batch, ConvLayerDesc, ThresholdsActivation
    rng = np.random.default_rng(5)
    K, C, D, O = 3, 32, 20, 16
    w = rng.integers(-2, 2, (O, K * K * C)).astype(np.int8)
    words = sicn_ref.pack_finn_tiles_generic(w, 8, 8, 2)
    desc = ConvLayerDesc(K=K, IFM_CH=C, IFM_DIM=D, OFM_CH=O, SIMD=8, PE=8, W_BIT=2, IN_SIGNED=False, OUT_BIT=8)
    thr = np.sort(rng.integers(-3000, 3000, (8, O // 8, 7)), axis=2).astype(np.int32)
    act = ThresholdsActivation(ACC_BIT=16, ACC_SIGNED=True, m_thresholds=thr, ACT_VAL=0)
    fpw = FixedPointWeights(8, 2, 8, desc.W_TILES, words)
    xs = [torch.from_numpy(rng.integers(0, 256, (2, D, D, C), dtype=np.uint8)).cuda() for _ in range(3)]
    want = [ConvLayer_Batch(desc, x, None, fpw, act, 2).clone() for x in xs]
    layer = ConvLayer(desc, fpw, act)
    got = [layer(x, None, 2) for x in xs]                  # three launches back to back, one handle
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        again = layer(xs[1], None, 2, stream=side)
    torch.cuda.synchronize()
    assert all(torch.equal(g, w_) for g, w_ in zip(got, want)) and torch.equal(again, want[1])
    layer.close()
    with pytest.raises(RuntimeError):
        layer(xs[0], None, 2)


# ---- sub-byte lanes (round 5): the streams carry IFM_CH * IN_BIT / OFM_CH * OUT_BIT bits per pixel -------------------
# (K, C, D, O, SIMD, PE, W_BIT, IN_BIT, IN_SIGNED, ACC_BIT, ACC_SIGNED, OUT_BIT, NUM_TH, ACT_VAL)
PACKED_CASES = [
    (3, 8, 9, 8, 4, 2, 4, 2, 0, 16, 1, 2, 3, 0),       # 2-bit unsigned lanes in, 3 thresholds -> 2-bit lanes out: a QNN inner layer
    (3, 16, 8, 4, 8, 2, 3, 4, 1, 12, 1, 4, 15, -8),    # 4-bit signed lanes in, 15 thresholds with ActVal -8 -> 4-bit two's-complement lanes
    (2, 32, 7, 16, 8, 4, 2, 1, 0, 8, 1, 2, 3, 0),      # binary lanes in (ap_uint<1>), 2-bit weights
    (1, 4, 6, 12, 2, 3, 5, 4, 0, 16, 1, 16, 0, 0),     # 4-bit lanes in, 16-bit pass-through containers out
    (3, 8, 10, 6, 2, 3, 4, 8, 0, 16, 1, 4, 7, 0),      # byte lanes in, 4-bit lanes out (6 lanes = 3 bytes per pixel)
    (5, 4, 9, 4, 4, 4, 4, 2, 1, 10, 1, 8, 0, 0),       # 2-bit SIGNED lanes (ap_int<2>: -2 .. 1) in, byte containers out
]


def _make_packed(case, rng):
    K, C, D, O, SIMD, PE, WB, IB, INS, AB, AS, OB, NTH, AV = case
    nf = O // PE
    w = rng.integers(-(1 << (WB - 1)), 1 << (WB - 1), (O, K * K * C)).astype(np.int8)
    words = sicn_ref.pack_finn_tiles_generic(w, SIMD, PE, WB)
    lanes = rng.integers(0, 1 << IB, (D, D, C)).astype(np.uint8)            # the lanes' bit patterns
    thr = None
    if NTH:
        lim = min(1 << (AB - 1), 60 * C)
        thr = np.sort(rng.integers(-lim if AS else 0, lim, (PE, nf, NTH)), axis=2).astype(np.int32)
    desc = SimpleNamespace(K=K, IFM_CH=C, IFM_DIM=D, OFM_CH=O, OFM_DIM=D - K + 1, SIMD=SIMD, PE=PE, IN_BIT=IB, IN_SIGNED=INS,
                           W_BIT=WB, W_TILES=nf * (K * K * C // SIMD), ACC_BIT=AB, ACC_SIGNED=AS, OUT_BIT=OB,
                           activation=int(NTH > 0), NUM_TH=NTH, ACT_VAL=AV)
    thr_oi = None if thr is None else thr.transpose(1, 0, 2).reshape(O, NTH)
    ref = sicn_ref.conv_layer_batch_ref(lanes, w, K, bool(INS), AB, bool(AS), OB, thr_oi, AV, in_bit=IB)
    return desc, w, words, lanes, thr, ref


@pytest.mark.parametrize("case", PACKED_CASES)
def test_dataflow_restatement_with_sub_byte_lanes_equals_closed_form(case):
    """TSrcI = Slice<ap_(u)int<IN_BIT>> with IN_BIT < 8 (interpret.hpp:191-244): the dataflow restatement (lanes travel one per byte
    through the sliding window, the MVAU reads their low IN_BIT bits, sign-extended for ap_int) against the closed form."""
    rng = np.random.default_rng(abs(hash(case)) % (1 << 32))
    desc, _, words, lanes, thr, ref = _make_packed(case, rng)
    for fsm in (True, False):
        assert np.array_equal(c_oracle.convlayer_dataflow(desc, words, thr, lanes, use_fsm=fsm), ref)
    junk = lanes | (rng.integers(0, 256, lanes.shape).astype(np.uint8) & np.uint8((0xFF << desc.IN_BIT) & 0xFF))   # bits above the lane are not the lane
    assert np.array_equal(c_oracle.convlayer_dataflow(desc, words, thr, junk, use_fsm=False), ref)


def test_stream_word_packing_is_the_reference_slice_layout():
    """pack_stream_lanes / unpack_stream_lanes: lane c of a pixel in bits [c W, (c + 1) W) of its word (interpret.hpp:191-244,
    convlayer.h:100), little-endian bytes — the layout conv3_nonsquare_tb.cpp:807-808 uses for 8-bit lanes, carried over to 1 / 2 / 4."""
    rng = np.random.default_rng(3)
    for bits, c in ((1, 16), (2, 8), (2, 12), (4, 6), (8, 5)):
        lanes = rng.integers(0, 1 << bits, (3, 4, c)).astype(np.uint8)
        words = sicn_ref.pack_stream_lanes(lanes, bits)
        assert words.shape == (3, 4, c * bits // 8) and words.dtype == np.uint8
        assert np.array_equal(sicn_ref.unpack_stream_lanes(words, bits, c), lanes)
        if bits == 8:
            assert np.array_equal(words, lanes)
    assert sicn_ref.pack_stream_lanes(np.array([[1, 2, 3, 0]], np.uint8), 2).tolist() == [[0b00111001]]       # lane 0 in the low bits
    assert sicn_ref.pack_stream_lanes(np.array([[0xA, 0x5]], np.uint8), 4).tolist() == [[0x5A]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", PACKED_CASES)
def test_gpu_conv_layer_batch_with_sub_byte_lanes_matches_oracle(case):
    """The GPU reads the packed input stream and writes the packed output stream (one thread per output byte); both sides of the
    comparison go through the independent Python restatement of the packing."""
    import torch
    from simple_image_compression_network_amd.api import FixedPointWeights
    from simple_image_compression_network_amd.convlayer import (ConvLayer_Batch, ConvLayerDesc, PassThroughActivation,
                                                                ThresholdsActivation)
    rng = np.random.default_rng(abs(hash(case)) % (1 << 32))
    K, C, D, O, SIMD, PE, WB, IB, INS, AB, AS, OB, NTH, AV = case
    d, _, words, lanes, thr, ref = _make_packed(case, rng)
    desc = ConvLayerDesc(K=K, IFM_CH=C, IFM_DIM=D, OFM_CH=O, SIMD=SIMD, PE=PE, W_BIT=WB, IN_SIGNED=bool(INS), OUT_BIT=OB, IN_BIT=IB)
    act = ThresholdsActivation(thr, AB, bool(AS), AV) if NTH else PassThroughActivation(AB, bool(AS))
    lanes2 = np.stack([lanes, lanes[::-1].copy()])
    xin = torch.from_numpy(sicn_ref.pack_stream_lanes(lanes2, IB)).cuda()
    out = ConvLayer_Batch(desc, xin, None, FixedPointWeights(SIMD, WB, PE, d.W_TILES, words), act, 2).cpu().numpy()
    got = sicn_ref.unpack_stream_lanes(out, OB, O) if OB < 8 else (out.astype(np.int64) & ((1 << OB) - 1))
    assert np.array_equal(got[0].astype(np.int64), ref.astype(np.int64))
    ref1 = c_oracle.convlayer_dataflow(d, words, thr, lanes[::-1].copy(), use_fsm=False)
    assert np.array_equal(got[1].astype(np.int64), ref1.astype(np.int64))


@pytest.mark.gpu
def test_gpu_two_thresholded_layers_chain_in_the_reference_packing():
    """8-bit lanes -> conv + 3 thresholds -> 2-bit lanes -> conv (IN_BIT = 2) + 15 thresholds -> 4-bit lanes.  The
    first layer's output BUFFER is handed to the second layer untouched: a ThresholdsActivation layer's output stream is byte-identical
    to the next layer's input stream (convlayer.h:100, activations.hpp:168-190).  Against the restated dataflow (oracle/), layer by
    layer and end to end."""
    import torch
    from simple_image_compression_network_amd.api import FixedPointWeights
    from simple_image_compression_network_amd.convlayer import ConvLayer_Batch, ConvLayerDesc, ThresholdsActivation
    rng = np.random.default_rng(2024)
    D, C0, C1, C2 = 14, 4, 16, 8
    x = rng.integers(0, 256, (2, D, D, C0), dtype=np.uint8)
    w1 = rng.integers(-8, 8, (C1, 9 * C0)).astype(np.int8)
    w2 = rng.integers(-2, 2, (C2, 9 * C1)).astype(np.int8)
    t1 = np.sort(rng.integers(-2000, 2000, (4, C1 // 4, 3)), axis=2).astype(np.int32)
    t2 = np.sort(rng.integers(-40, 40, (2, C2 // 2, 15)), axis=2).astype(np.int32)
    d1 = ConvLayerDesc(K=3, IFM_CH=C0, IFM_DIM=D, OFM_CH=C1, SIMD=4, PE=4, W_BIT=4, IN_SIGNED=False, OUT_BIT=2)
    d2 = ConvLayerDesc(K=3, IFM_CH=C1, IFM_DIM=D - 2, OFM_CH=C2, SIMD=8, PE=2, W_BIT=2, IN_SIGNED=False, OUT_BIT=4, IN_BIT=2)
    a1, a2 = ThresholdsActivation(t1, 16, True, 0), ThresholdsActivation(t2, 12, True, 0)
    f1 = FixedPointWeights(4, 4, 4, d1.W_TILES, sicn_ref.pack_finn_tiles_generic(w1, 4, 4, 4))
    f2 = FixedPointWeights(8, 2, 2, d2.W_TILES, sicn_ref.pack_finn_tiles_generic(w2, 8, 2, 2))
    mid = ConvLayer_Batch(d1, torch.from_numpy(x).cuda(), None, f1, a1, 2)
    assert tuple(mid.shape) == (2, D - 2, D - 2, C1 * 2 // 8)                  # 16 lanes x 2 bit = 4 bytes per pixel
    out = ConvLayer_Batch(d2, mid, None, f2, a2, 2)                             # the SAME buffer, no repacking
    assert tuple(out.shape) == (2, D - 4, D - 4, C2 * 4 // 8)

    def odesc(d, act):
        return SimpleNamespace(K=d.K, IFM_CH=d.IFM_CH, IFM_DIM=d.IFM_DIM, OFM_CH=d.OFM_CH, OFM_DIM=d.OFM_DIM, SIMD=d.SIMD, PE=d.PE,
                               IN_BIT=d.IN_BIT, IN_SIGNED=0, W_BIT=d.W_BIT, W_TILES=d.W_TILES, ACC_BIT=act.ACC_BIT, ACC_SIGNED=1,
                               OUT_BIT=d.OUT_BIT, activation=1, NUM_TH=act.m_thresholds.shape[2], ACT_VAL=0)
    for i in range(2):
        m = c_oracle.convlayer_dataflow(odesc(d1, a1), f1.m_weights, t1, x[i], use_fsm=True)          # uint32 lanes, 0 .. 3
        assert np.array_equal(sicn_ref.unpack_stream_lanes(mid[i].cpu().numpy(), 2, C1), m)
        assert np.array_equal(mid[i].cpu().numpy(), sicn_ref.pack_stream_lanes(m, 2))                  # byte-identical stream
        o = c_oracle.convlayer_dataflow(odesc(d2, a2), f2.m_weights, t2, m.astype(np.uint8), use_fsm=True)
        assert np.array_equal(out[i].cpu().numpy(), sicn_ref.pack_stream_lanes(o, 4))
