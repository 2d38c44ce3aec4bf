"""Fixed-point GDN / IGDN, specification version 2 (include/sicn_gdn.h).  NEW functionality without a
reference counterpart (activations.hpp:127-224 has no GDN), parity "unpinned": the CPU tests hold the two independent statements
of the specification together (oracle/sicn_gdn_oracle.c: 128-bit bisection over the 2048 classes of nq + the C library's fmaf /
nearbyintf; oracle/sicn_ref.py: Python integers only), re-derive the safety margin of the rounding biases from exact integers and
check the defining properties; the GPU tests show the HIP kernels (MFMA cross-channel sum, one hardware root + one multiply + two
masks, fma, saturating conversion) reproduce it bit for bit — standalone, behind every layer kernel family, in every internal
layout of a chain — and that the hardware root is exact for EVERY n."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import c_oracle, sicn_ref
from simple_image_compression_network_amd.config import LayerDesc

ROOT = Path(__file__).resolve().parent.parent
gpu = pytest.mark.gpu


def _params(rng, c, kind="random"):
    if kind == "unit":        # beta = 1.0 (Q8), gamma = 0: y = x * 2^(16-8/2... ) pure scaling
        return np.full(c, 256, np.uint32), np.zeros((c, c), np.uint8)
    if kind == "extreme":     # largest n the specification allows
        return np.full(c, 65535, np.uint32), np.full((c, c), 127, np.uint8)
    beta = rng.integers(1, 65536, c).astype(np.uint32)
    gamma = rng.integers(0, 128, (c, c)).astype(np.uint8)
    gamma[rng.random((c, c)) < 0.5] = 0
    return beta, gamma


@pytest.mark.parametrize("c", [1, 3, 6, 64, 128, 192])
@pytest.mark.parametrize("inverse", [False, True])
def test_two_oracle_statements_agree(c, inverse):
    rng = np.random.default_rng(c * 2 + inverse)
    x = rng.integers(0, 256, (40, c), dtype=np.uint8)
    x[0, :] = 0x80                       # -128 (version 1 clamped it; version 2 takes it as it is: x^2 = 16384)
    x[1, :] = 0x7F
    x[2, :] = 0
    for kind in ("random", "unit", "extreme"):
        beta, gamma = _params(rng, c, kind)
        for shift in (1, 8, 12, 24):
            a = c_oracle.gdn(x, beta, gamma, inverse, shift)
            assert np.array_equal(a, sicn_ref.gdn_ref(x, beta, gamma, inverse, shift)), (kind, shift)


def test_two_oracle_statements_agree_on_dense_single_channel_sweeps():
    """One channel: n = beta + gamma x^2 exactly, so beta sweeps n through every small value, the powers of two and their
    neighbours (where nq's exponent and parity change), and all 256 lanes meet every root — for every shift."""
    rng = np.random.default_rng(11)
    xs = np.arange(256, dtype=np.uint8).reshape(256, 1)
    edges = np.concatenate([[(1 << k) - 1, 1 << k, (1 << k) + 1] for k in range(1, 16)])
    betas = np.unique(np.clip(np.concatenate([np.arange(1, 2200), edges, np.arange(65000, 65536), rng.integers(1, 65536, 400)]), 1, 65535))
    for inverse in (False, True):
        for shift in range(1, 25):
            for beta in betas[:: (1 if shift in (8, 12, 16) else 37)]:
                for gam in (0, 127):
                    b, gm = np.array([beta], np.uint32), np.array([[gam]], np.uint8)
                    assert np.array_equal(c_oracle.gdn(xs, b, gm, inverse, shift), sicn_ref.gdn_ref(xs, b, gm, inverse, shift)), (inverse, shift, int(beta), gam)


@pytest.mark.parametrize("inverse", [False, True])
def test_rounding_bias_keeps_every_class_away_from_a_step(inverse):
    """The reason the specification can be computed with a 1-ulp hardware root: for each of the 2 x 1024 values (exponent parity, 10
    fraction bits) nq can take, root(nq) (1 + b 2^-16) lies >= 7.5 binary32 ulps from the next multiple of 2^-10 (relative): an error
    of 1 ulp in the root plus 1/2 ulp in the multiply cannot move trunc11 to the other side.  Exact integers (isqrt at 2^-64)."""
    import math
    b = sicn_ref.GDN_BIAS[1 if inverse else 0]
    worst = None
    for par in range(2):
        for frac in range(1024):
            m = (1024 + frac) << par                                          # nq 2^10
            num, den = ((65536 + b) ** 2 * m, 1 << 10) if inverse else ((65536 + b) ** 2 << 10, m)   # V^2 2^32
            v = math.isqrt((num << 160) // den)                               # V 2^96, floor
            top = v >> (v.bit_length() - 40)                                  # 40 significant bits: 2^16 units per binary32 ulp
            pos = top & ((1 << 29) - 1)                                       # position inside a step of trunc11 (2^13 ulps)
            margin = min(pos, (1 << 29) - pos) / 65536.0
            worst = margin if worst is None else min(worst, margin)
    assert worst >= 7.5, worst


def _roots_as_m_k(r: np.ndarray):
    """binary32 roots -> (M, k) with r = M 2^k, 1024 <= M < 2048 (exact: r has 11 significant bits)."""
    b = np.ascontiguousarray(r, np.float32).view(np.uint32).astype(np.int64)
    assert np.all((b & 0x1FFF) == 0) and np.all(b >> 31 == 0)
    return ((b & 0x7FFFFF) | 0x800000) >> 13, (b >> 23) - 127 - 10


@pytest.mark.parametrize("inverse", [False, True])
def test_two_oracle_statements_agree_on_every_root(inverse):
    """EXHAUSTIVE over what r can depend on: every class of nq (2 parities x 1024 fractions) at every exponent a
    u32 n can have, at the class's first and last n and their neighbours (where the 24-bit rounding of n >= 2^24 and the cut to 11 bits
    decide), for every shift 1 .. 24 — the C statement (binary32 conversion, 128-bit bisection over the classes, ldexp) against the
    Python one (integer rounding, math.isqrt on big integers)."""
    ns = set()
    for e in range(0, 31):                                           # n in [2^e, 2^(e+1))
        lo = 1 << e
        width = max(lo >> 10, 1)                                     # n's per class (>= 1)
        for f in range(0, 1024) if e >= 10 else range(0, lo):
            first = lo + f * width
            for n in (first - 1, first, first + 1, first + width // 2 - 1, first + width // 2, first + width - 1):
                if 1 <= n < (1 << 31):
                    ns.add(n)
    ns = np.array(sorted(ns), np.uint32)
    assert ns.size > 100000
    quant = [sicn_ref.gdn_quantise_n(int(n)) for n in ns]
    for shift in range(1, 25):
        r = c_oracle.gdn_roots(ns, inverse, shift)
        big_m, k = _roots_as_m_k(r)
        # the mantissa of r does not depend on the shift (trunc11 is scale-invariant): the full comparison at every shift for a stride
        # of the n's, and for ALL of them at three shifts
        step = 1 if shift in (1, 12, 24) else 41
        for i in range(0, ns.size, step):
            assert (int(big_m[i]), int(k[i])) == sicn_ref.gdn_root(*quant[i], inverse, shift), (int(ns[i]), shift)
        if shift > 1:      # ... and the scale-invariance itself, for every n: one more shift = one less in the exponent, the same mantissa
            assert np.array_equal(big_m, prev_m) and np.array_equal(k, prev_k - 1), shift
        prev_m, prev_k = big_m, k


def test_two_oracle_statements_agree_on_every_output():
    """EXHAUSTIVE over the output step: every 11-bit root mantissa (1024) x every exponent the roots can have (k = -34 .. 13: both
    activations, all shifts, all n) x every lane value (256) — 12.6 M combinations: fmaf + nearbyintf + clamp in C against the two
    integer roundings in Python."""
    m = np.arange(1024, 2048, dtype=np.int64)
    x = np.arange(-128, 128, dtype=np.int64)
    for k in range(-34, 14):
        r = np.ldexp(m.astype(np.float64), k).astype(np.float32)                 # exact: 11 significant bits
        rr, xx = np.meshgrid(r, x.astype(np.int8), indexing="ij")
        got = c_oracle.gdn_outputs(xx, rr)
        want = sicn_ref.gdn_output_ref(x[None, :], m[:, None], np.int64(k))
        assert np.array_equal(got, want), k


def test_specification_properties():
    rng = np.random.default_rng(5)
    c = 16
    x = rng.integers(0, 256, (200, c), dtype=np.uint8)
    x[0, :4] = 0x80
    beta, gamma = _params(rng, c, "unit")
    # beta = 256 (1.0 in Q8), gamma = 0, shift = 12: GDN r = trunc11(2^4 (1 + 5 2^-16) / 16) = 1 -> y = x exactly (identity, -128 included)
    y = c_oracle.gdn(x, beta, gamma, False, 12)
    assert np.array_equal(y, x)
    # IGDN with the same parameters: r = trunc11(2^-4 (1 + 33 2^-16) 16) = 1 -> identity as well
    y = c_oracle.gdn(x, beta, gamma, True, 12)
    assert np.array_equal(y, x)
    # odd symmetry: y(-x) = -y(x) up to the tie rule (nearest-even of u = x r + 128 on binary32's grid, finer below 128 than above)
    beta, gamma = _params(rng, c)
    xs = np.maximum(x.view(np.int8), -127)
    yp = c_oracle.gdn(xs.view(np.uint8), beta, gamma, False, 10).view(np.int8).astype(int)
    yn = c_oracle.gdn((-xs).astype(np.int8).view(np.uint8), beta, gamma, False, 10).view(np.int8).astype(int)
    assert np.all(np.abs(yp + yn) <= 1)
    # GDN divides: more energy in the other channels never increases |y|
    g2 = gamma.copy()
    g2[g2 < 127] += 1
    y1 = np.abs(c_oracle.gdn(x, beta, gamma, False, 10).view(np.int8).astype(int))
    y2 = np.abs(c_oracle.gdn(x, beta, g2, False, 10).view(np.int8).astype(int))
    assert np.all(y2 <= y1)
    # rejected parameters
    with pytest.raises(RuntimeError):
        c_oracle.gdn(x, np.zeros(c, np.uint32), gamma, False, 12)          # beta = 0
    with pytest.raises(RuntimeError):
        c_oracle.gdn(x, beta, np.full((c, c), 128, np.uint8), False, 12)   # gamma > 127


def test_preact_lane_is_the_reference_layer_without_its_relu():
    """relu7(pre-activation) == the reference layer: the GDN input is exactly conv_nonsquare_top.cpp:272's lane."""
    rng = np.random.default_rng(9)
    d = LayerDesc(IFM_CH=6, IFM_ROW=11, IFM_COL=7, OFM_CH=4, OFM_ROW=6, OFM_COL=4, SIMD=3, PE=2, W_TILES=2 * 50)
    W = rng.integers(-8, 8, (4, 5, 5, 6)).astype(np.int8)
    b = rng.integers(-128, 128, 4).astype(np.int8)
    x = rng.integers(0, 256, (7, 11, 6), dtype=np.uint8)
    words = sicn_ref.pack_finn_tiles(W, 3, 2)
    pre = c_oracle.run_layer_preact(d, words, b, x)
    assert np.array_equal(pre, sicn_ref.layer_preact_ref(x, W, b, 0))
    relu = pre.copy()
    relu[relu >= 128] = 0
    assert np.array_equal(relu, c_oracle.run_layer(d, words, b, x, "direct"))
    assert np.array_equal(relu, sicn_ref.conv2d_ref(x, W, b))


def test_gdn_abi_symbols_exported():
    text = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "sicn_gdn.h").read_text(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(sicn_[a-z0-9_]+)\s*\(", text)))
    from simple_image_compression_network_amd import _lib
    L = _lib.lib()
    assert set(syms) == set(_lib.GDN_ABI) and all(hasattr(L, s) for s in syms)
    out = ctypes.c_void_p()
    beta = (ctypes.c_uint32 * 4)(1, 2, 3, 0)          # beta = 0 is rejected on the host, before any GPU call
    gamma = (ctypes.c_uint8 * 16)()
    assert L.sicn_gdn_create(4, 0, 12, beta, gamma, ctypes.byref(out)) == -22
    assert L.sicn_gdn_create(4, 2, 12, beta, gamma, ctypes.byref(out)) == -22
    assert L.sicn_gdn_create(4, 0, 0, beta, gamma, ctypes.byref(out)) == -22
    assert L.sicn_gdn_apply(None, None, 4, None) == -22


# ------------------------------------------------------------------------------------------ GPU
def _mk_desc(cin, cout, simd, pe, w, h, tr):
    ow, oh = (2 * w, 2 * h) if tr else ((w + 1) // 2, (h + 1) // 2)
    d = LayerDesc(IFM_CH=cin, IFM_ROW=w, IFM_COL=h, OFM_CH=cout, OFM_ROW=ow, OFM_COL=oh, SIMD=simd, PE=pe,
                  W_TILES=(cout // pe) * (25 * cin // simd), transposed=tr)
    d.validate()
    return d


@gpu
@pytest.mark.parametrize("c", [128, 192, 6, 64, 3])
@pytest.mark.parametrize("inverse", [False, True])
def test_gpu_gdn_apply_equals_oracle(c, inverse):
    import torch
    from simple_image_compression_network_amd import api
    rng = np.random.default_rng(c + inverse)
    npos = 1000 if c >= 64 else 333                    # not a multiple of the 256-position block
    x = rng.integers(0, 256, (npos, c), dtype=np.uint8)
    x[0, :] = 0x80
    for kind, shift in (("random", 12), ("extreme", 16), ("unit", 12), ("random", 1), ("random", 24)):
        beta, gamma = _params(rng, c, kind)
        g = api.GDN(beta, gamma, inverse, shift)
        got = g.apply_(torch.from_numpy(x).cuda()).cpu().numpy()
        ref = c_oracle.gdn(x, beta, gamma, inverse, shift)
        assert np.array_equal(got, ref), (kind, shift, np.count_nonzero(got != ref))


@gpu
def test_gpu_gdn_square_roots_exhaustive_ranges():
    """Sweep n densely through small values, perfect squares +-1 and the top of the range with a 1-channel activation whose n is
    exactly beta + gamma * x^2 (the generic kernel: the same gdn_root / fma / conversion as the MFMA kernels)."""
    import torch
    from simple_image_compression_network_amd import api
    rng = np.random.default_rng(1)
    xs = np.arange(256, dtype=np.uint8).reshape(256, 1)
    betas = np.unique(np.concatenate([np.arange(1, 600), np.arange(16000, 16700), np.arange(65000, 65536),
                                      (np.arange(2, 256) ** 2), (np.arange(2, 256) ** 2) - 1, (np.arange(2, 256) ** 2) + 1,
                                      rng.integers(1, 65536, 300)]))
    for inverse in (False, True):
        for gam in (0, 1, 127):
            for beta in betas[:: 7 if gam else 1]:
                b = np.array([beta], np.uint32)
                gm = np.array([[gam]], np.uint8)
                g = api.GDN(b, gm, inverse, 9)
                got = g.apply_(torch.from_numpy(xs).cuda()).cpu().numpy()
                assert np.array_equal(got, c_oracle.gdn(xs, b, gm, inverse, 9)), (inverse, gam, int(beta))


@gpu
@pytest.mark.parametrize("inverse", [0, 1])
def test_gpu_gdn_roots_exact_for_every_n(inverse):
    """sicn_gdn_selftest_roots: the root exactly as the kernels compute it — v_cvt_f32_u32, a mask, v_rsq_f32 / v_sqrt_f32 (1 ulp), one
    multiply, a mask — against the defining integer inequalities ON THE DEVICE for EVERY n the specification admits (n < 2^31; the
    MFMA kernels see n < 2^29, the generic one n < 2^16 + 1024 * 127 * 16384 < 2^31).  0 mismatches = the hardware root never lands on
    the wrong side of a step of trunc11, for any exponent — not merely on samples."""
    from simple_image_compression_network_amd import _lib
    L = _lib.lib()
    assert L.sicn_gdn_spec_version() == 2
    step = 1 << 28
    for begin in range(0, 1 << 31, step):
        assert L.sicn_gdn_selftest_roots(inverse, begin, step) == 0, (inverse, begin)
    assert L.sicn_gdn_selftest_roots(inverse, 0, (1 << 31) + 1) == -22


# every kernel family that can carry a GDN: l0_rgb, mfma_conv (128 and 192 out), mfma_deconv, generic (incl. the RGB-out
# layer, which the library routes to the generic kernel when it has a GDN)
GDN_LAYERS = [(3, 128, 3, 8, 70, 38, 0), (128, 128, 8, 16, 66, 18, 0), (128, 192, 8, 24, 40, 22, 0), (192, 128, 12, 16, 33, 9, 1),
              (128, 128, 8, 16, 34, 10, 1), (128, 3, 8, 3, 20, 7, 1), (6, 6, 2, 3, 13, 9, 0), (192, 128, 12, 16, 37, 21, 0),
              (128, 192, 8, 24, 19, 11, 1)]


@gpu
@pytest.mark.parametrize("case", GDN_LAYERS)
@pytest.mark.parametrize("inverse", [False, True])
def test_gpu_layer_with_gdn_equals_oracle(case, inverse):
    import torch
    from simple_image_compression_network_amd import api
    rng = np.random.default_rng(abs(hash(case)) % (1 << 31) + inverse)
    d = _mk_desc(*case)
    W = rng.integers(-8, 8, (d.OFM_CH, 5, 5, d.IFM_CH)).astype(np.int8)
    b = rng.integers(-128, 128, d.OFM_CH).astype(np.int8)
    words = sicn_ref.pack_finn_tiles(W, d.SIMD, d.PE)
    x = rng.integers(0, 256, (2,) + d.in_shape, dtype=np.uint8)
    beta, gamma = _params(rng, d.OFM_CH)
    g = api.GDN(beta, gamma, inverse, 12)
    fpw = api.FixedPointWeights(d.SIMD, 4, d.PE, d.W_TILES, words)
    fn = api.deconv522 if d.transposed else api.conv2d
    got = fn(d, fpw, b, torch.from_numpy(x).cuda(), None, 2, gdn=g).cpu().numpy()
    for i in range(2):
        pre = sicn_ref.layer_preact_ref(x[i], W, b, d.transposed)
        assert np.array_equal(got[i], c_oracle.gdn(pre, beta, gamma, inverse, 12)), i


@gpu
@pytest.mark.parametrize("grid", [0, 8, 16])
@pytest.mark.parametrize("case", [(128, 128, 8, 16, 66, 18, 0), (128, 128, 8, 16, 34, 10, 1), (128, 128, 8, 16, 131, 33, 0), (128, 128, 8, 16, 65, 19, 1)])
@pytest.mark.parametrize("inverse", [False, True])
def test_gpu_wide_persistent_layer_with_gdn_equals_oracle(case, inverse, grid):
    """the wide persistent kernels (k_conv_x / k_deconv_x, chosen automatically from 4 tiles per CU on — i.e. for the
    hyperprior's main transform at 8 x 4K) hand their accumulators over with the RAW floor when a GDN follows; that path had no
    oracle comparison (the GDN test sizes never reached the automatic threshold).  Forced here (wave_tile = 128) with 8 / 16 / all
    workgroups so that every workgroup walks several tiles."""
    import torch
    from simple_image_compression_network_amd import api
    rng = np.random.default_rng(abs(hash(case)) % (1 << 31) + 7 * inverse + grid)
    d = _mk_desc(*case)
    W = rng.integers(-8, 8, (d.OFM_CH, 5, 5, d.IFM_CH)).astype(np.int8)
    b = rng.integers(-128, 128, d.OFM_CH).astype(np.int8)
    words = sicn_ref.pack_finn_tiles(W, d.SIMD, d.PE)
    x = rng.integers(0, 256, (2,) + d.in_shape, dtype=np.uint8)
    beta, gamma = _params(rng, d.OFM_CH)
    g = api.GDN(beta, gamma, inverse, 12)
    fpw = api.FixedPointWeights(d.SIMD, 4, d.PE, d.W_TILES, words)
    fn = api.deconv522 if d.transposed else api.conv2d
    got = fn(d, fpw, b, torch.from_numpy(x).cuda(), None, 2, gdn=g, options={"wave_tile": 128, "persistent_grid": grid}).cpu().numpy()
    for i in range(2):
        pre = sicn_ref.layer_preact_ref(x[i], W, b, d.transposed)
        assert np.array_equal(got[i], c_oracle.gdn(pre, beta, gamma, inverse, 12)), i


@gpu
@pytest.mark.parametrize("options", [{}, pytest.param({"gdn_fuse": 2}, marks=pytest.mark.alt), {"gdn_fuse": 1},
                                     pytest.param({"gdn_fuse": 2, "no_phase_layout": 1}, marks=pytest.mark.alt), {"tile_x": 16}, {"no_phase_layout": 1},
                                     {"force_generic": 1}, {"wave_tile": 128, "persistent_grid": 8}])
def test_gpu_gdn_net_all_internal_layouts(options):
    """The hyperprior-style main transform: GDN after L0-L2, IGDN after L4-L6 (L3 and L7 keep the reference's ReLU), as one
    sicn_net chain.  The activations run in place on GROUP / PHASE / NHWC intermediates; latent and reconstruction must
    equal the oracle's layer-by-layer statement."""
    import torch
    from simple_image_compression_network_amd import api
    from simple_image_compression_network_amd.config import eight_layer_descs
    rng = np.random.default_rng(42)
    w, h = (96, 64) if not options.get("force_generic") else (32, 16)
    descs = eight_layer_descs(w, h)
    params_np, params, gdns, gdn_np = [], [], [], []
    for l, d in enumerate(descs):
        Wt = rng.integers(-8, 8, (d.OFM_CH, 5, 5, d.IFM_CH)).astype(np.int8)
        bt = rng.integers(-128, 128, d.OFM_CH).astype(np.int8)
        params_np.append((Wt, bt, d.transposed))
        params.append((api.FixedPointWeights(d.SIMD, 4, d.PE, d.W_TILES, sicn_ref.pack_finn_tiles(Wt, d.SIMD, d.PE)),
                       api.FixedPointWeights(1, 8, 1, d.OFM_CH, bt.view(np.uint8).astype(np.uint64))))
        if l in (0, 1, 2, 4, 5, 6):
            beta, gamma = _params(rng, d.OFM_CH)
            gdn_np.append((beta, gamma, l >= 4))
            gdns.append(api.GDN(beta, gamma, l >= 4, 12))
        else:
            gdn_np.append(None)
            gdns.append(None)
    net = api.EightLayersNet(w, h, params=params, gdn=gdns, options=options or None)
    x = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    out, lat = net.forward(torch.from_numpy(x).cuda())
    out, lat = out.cpu().numpy(), lat.cpu().numpy()
    for i in range(2):
        a = x[i]
        for l, (Wt, bt, tr) in enumerate(params_np):
            if gdn_np[l] is None:
                a = sicn_ref.deconv522_ref(a, Wt, bt) if tr else sicn_ref.conv2d_ref(a, Wt, bt)
            else:
                beta, gamma, inv = gdn_np[l]
                a = c_oracle.gdn(sicn_ref.layer_preact_ref(a, Wt, bt, tr), beta, gamma, inv, 12)
            if l == 3:
                assert np.array_equal(lat[i], a), "latent"
        assert np.array_equal(out[i], a), "reconstruction"


@gpu
@pytest.mark.parametrize("size", [(70, 38), (5, 3), (129, 17), (64, 2), (1, 1), (140, 150), (513, 31), (70, 290)])
@pytest.mark.parametrize("inverse", [False, True])
def test_gpu_layer0_with_gdn_in_one_kernel_equals_oracle(size, inverse):
    """k_l0g (round 4): layer 0 and its GDN / IGDN in one kernel — the pre-activation tensor stays in registers.  Sizes: sub-tile,
    odd, one row / one pixel, several 32-column strips, runs of 4 tiles with a shorter last run (70 x 290: 19 tiles), pixels
    >= 128; three images.  Held to the oracle AND to the two-kernel path (sicn_options.gdn_fuse = 1), with the default cut and
    with the strips forced into 1 / 3 / 5 runs."""
    import torch
    from simple_image_compression_network_amd import api
    rng = np.random.default_rng(abs(hash(size)) % (1 << 31) + inverse)
    d = _mk_desc(3, 128, 3, 8, size[0], size[1], 0)
    W = rng.integers(-8, 8, (128, 5, 5, 3)).astype(np.int8)
    b = rng.integers(-128, 128, 128).astype(np.int8)
    words = sicn_ref.pack_finn_tiles(W, d.SIMD, d.PE)
    x = rng.integers(0, 256, (3,) + d.in_shape, dtype=np.uint8)
    beta, gamma = _params(rng, 128)
    g = api.GDN(beta, gamma, inverse, 12)
    fpw = api.FixedPointWeights(d.SIMD, 4, d.PE, d.W_TILES, words)
    xd = torch.from_numpy(x).cuda()
    got = api.conv2d(d, fpw, b, xd, None, 3, gdn=g).cpu().numpy()
    for i in range(3):
        pre = sicn_ref.layer_preact_ref(x[i], W, b, 0)
        assert np.array_equal(got[i], c_oracle.gdn(pre, beta, gamma, inverse, 12)), i
    assert np.array_equal(got, api.conv2d(d, fpw, b, xd, None, 3, gdn=g, options={"gdn_fuse": 1}).cpu().numpy())
    for chunks in (1, 3, 5):
        assert np.array_equal(got, api.conv2d(d, fpw, b, xd, None, 3, gdn=g, options={"strip_chunks": chunks}).cpu().numpy()), chunks


@gpu
def test_gpu_layer0_with_gdn_in_one_kernel_at_1080p():
    """Two workgroups on every CU (2 x 1080p = 4080 workgroups of 512 threads), the layouts of a chain: against the two-kernel path."""
    import torch
    from simple_image_compression_network_amd import api
    rng = np.random.default_rng(31)
    d = _mk_desc(3, 128, 3, 8, 1920, 1080, 0)
    W = rng.integers(-8, 8, (128, 5, 5, 3)).astype(np.int8)
    b = rng.integers(-128, 128, 128).astype(np.int8)
    fpw = api.FixedPointWeights(d.SIMD, 4, d.PE, d.W_TILES, sicn_ref.pack_finn_tiles(W, d.SIMD, d.PE))
    xd = torch.from_numpy(rng.integers(0, 256, (2,) + d.in_shape, dtype=np.uint8)).cuda()
    beta, gamma = _params(rng, 128)
    g = api.GDN(beta, gamma, False, 12)
    a = api.conv2d(d, fpw, b, xd, None, 2, gdn=g)
    assert torch.equal(a, api.conv2d(d, fpw, b, xd, None, 2, gdn=g, options={"gdn_fuse": 1}))


def _rgb_tail_chain(rng, first_case, inverse, options={"gdn_fuse": 2}):
    """(a 128-channel layer with a GDN / IGDN) -> (128 channels -> RGB deconv): the tail of the hyperprior's synthesis."""
    from simple_image_compression_network_amd import api
    d0 = _mk_desc(*first_case)
    d1 = _mk_desc(128, 3, 8, 3, d0.OFM_ROW, d0.OFM_COL, 1)
    layers = []
    for d in (d0, d1):
        Wt = rng.integers(-8, 8, (d.OFM_CH, 5, 5, d.IFM_CH)).astype(np.int8)
        bt = rng.integers(-128, 128, d.OFM_CH).astype(np.int8)
        layers.append((Wt, bt, d.transposed))
    params = [(api.FixedPointWeights(d.SIMD, 4, d.PE, d.W_TILES, sicn_ref.pack_finn_tiles(Wt, d.SIMD, d.PE)),
               api.FixedPointWeights(1, 8, 1, d.OFM_CH, bt.view(np.uint8).astype(np.uint64))) for d, (Wt, bt, _) in zip((d0, d1), layers)]
    beta, gamma = _params(rng, 128)
    net = api.EightLayersNet(descs=[d0, d1], params=params, gdn=[api.GDN(beta, gamma, inverse, 12), None], options=options)
    return net, d0, layers, (beta, gamma)


# first layer: (cin, cout, simd, pe, w, h, transposed) — its output is the RGB layer's input: 2 x 2 (one position per side), 16 x 10,
# 34 x 18, 31 x 18 (odd width: byte stores), 62 x 40 (exactly one strip of 62 columns, ten steps of 4 rows), 90 x 66 (two strips:
# 62 + 28), 126 x 14 (three strips: 62 + 62 + 2), 63 x 9 (conv: a strip of one column, odd height)
RGB_TAIL = [(128, 128, 8, 16, 1, 1, 1), (128, 128, 8, 16, 8, 5, 1), (128, 128, 8, 16, 17, 9, 1), (128, 128, 8, 16, 61, 35, 0),
            (128, 128, 8, 16, 31, 20, 1), (128, 128, 8, 16, 45, 33, 1), (128, 128, 8, 16, 63, 7, 1), (128, 128, 8, 16, 125, 17, 0)]


@pytest.mark.alt
@gpu
@pytest.mark.parametrize("case", RGB_TAIL)
@pytest.mark.parametrize("inverse", [False, True])
def test_gpu_rgb_layer_applies_previous_gdn_equals_oracle(case, inverse):
    """k_l7g (round 4, sicn_options.gdn_fuse = 2): in a chain the 128 -> RGB layer reads the PRE-activation lanes of the layer before it
    and applies that layer's GDN / IGDN on the way into its LDS window.  Against the oracle, against the three-kernel path (gdn_fuse = 0 / 1), with
    the strips cut into 1 / 2 / 3 runs, and with the first layer tapped (its activated output is then wanted: no fusion)."""
    import torch
    rng = np.random.default_rng(abs(hash(case)) % (1 << 31) + 3 * inverse)
    net, d0, layers, (beta, gamma) = _rgb_tail_chain(rng, case, inverse)
    x = rng.integers(0, 128, (2,) + d0.in_shape, dtype=np.uint8)
    xd = torch.from_numpy(x).cuda()
    got = net.run_layers(0, 1, xd)[0].cpu().numpy()
    for i in range(2):
        pre = sicn_ref.layer_preact_ref(x[i], layers[0][0], layers[0][1], layers[0][2])
        mid = c_oracle.gdn(pre, beta, gamma, inverse, 12)
        assert np.array_equal(got[i], sicn_ref.deconv522_ref(mid, layers[1][0], layers[1][1])), i
    for opt in ({"gdn_fuse": 1}, {"gdn_fuse": 0}, {"strip_chunks": 1}, {"strip_chunks": 2}, {"strip_chunks": 3}, {"strip_chunks": 7}, {"no_phase_layout": 1}, {"tile_x": 16}):
        opt = dict({"gdn_fuse": 2}, **opt)
        rng2 = np.random.default_rng(abs(hash(case)) % (1 << 31) + 3 * inverse)
        net2 = _rgb_tail_chain(rng2, case, inverse, opt)[0]
        assert np.array_equal(got, net2.run_layers(0, 1, xd)[0].cpu().numpy()), opt
    out, tap = net.run_layers(0, 1, xd, tap_layer=0)
    assert np.array_equal(out.cpu().numpy(), got)
    pre = sicn_ref.layer_preact_ref(x[1], layers[0][0], layers[0][1], layers[0][2])
    assert np.array_equal(tap[1].cpu().numpy(), c_oracle.gdn(pre, beta, gamma, inverse, 12))


@pytest.mark.alt
@gpu
def test_gpu_rgb_layer_applies_previous_gdn_at_1080p_input():
    """One workgroup of 1024 threads on every CU, 16 strips x 135 steps cut by sicn_plan.h: 960 x 540 -> (IGDN) -> 1920 x 1080 x 128 -> RGB 3840 x 2160,
    against the three-kernel path."""
    import torch
    rng = np.random.default_rng(9)
    net, d0, _, _ = _rgb_tail_chain(rng, (128, 128, 8, 16, 960, 540, 1), True)
    rng = np.random.default_rng(9)
    ref = _rgb_tail_chain(rng, (128, 128, 8, 16, 960, 540, 1), True, {"gdn_fuse": 1})[0]
    xd = torch.from_numpy(np.random.default_rng(1).integers(0, 128, (1,) + d0.in_shape, dtype=np.uint8)).cuda()
    assert torch.equal(net.run_layers(0, 1, xd)[0], ref.run_layers(0, 1, xd)[0])
