#!/usr/bin/env python3
"""Randomised parity sweep: every kernel family on random image sizes / batch sizes / strip cuts / tile widths against
the numpy closed form (oracle/sicn_ref.py), bit for bit.  usage: fuzz_parity.py [--cases N] [--seed S]"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import sicn_ref  # noqa: E402  (checker)
from simple_image_compression_network_amd import api  # noqa: E402
from simple_image_compression_network_amd.config import LayerDesc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--big", action="store_true", help="image sizes up to 600 x 400 (slow on the CPU side)")
ap.add_argument("--chains", type=int, default=0, help="additionally: whole 8-layer chains (internal layouts) on random sizes")
ap.add_argument("--gdn", type=int, default=0, help="additionally: layer 0 + GDN in one kernel (k_l0g) and (layer + GDN) -> RGB layer chains with the "
                "activation applied by the RGB layer (k_l7g), random sizes, against the C oracle of the GDN")
ap.add_argument("--deal", type=int, default=0, help="additionally: one-layer nets of conv / deconv 128 -> 128 on few wide persistent workgroups that walk >= 16 "
                "tiles each, so that the dynamic part of the tile deal (k_mfma16x.hip DealX: tickets, stealing across XCDs) is what runs; three "
                "calls + a graph replay each")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)

FAMILIES = [  # (cin, cout, simd, pe, transposed)
    (3, 128, 3, 8, 0), (128, 128, 8, 16, 0), (128, 192, 8, 24, 0), (192, 128, 12, 16, 1), (128, 128, 8, 16, 1), (128, 3, 8, 3, 1)]
HAS_ALT = api._lib.lib().sicn_has_alt_kernels() == 1   # SICN_LIB=.../libsicn_alt.so: also draws the forms that live in the ALT build only
bad = 0
for case in range(args.cases):
    cin, cout, simd, pe, tr = FAMILIES[rng.integers(len(FAMILIES))]
    big = rng.random() < 0.25
    w = int(rng.integers(1, 600 if args.big else 200 if big else 70))
    h = int(rng.integers(1, 400 if args.big else 120 if big else 40))
    if cin == 3:
        w, h = w * 2 + int(rng.integers(2)), h * 2 + int(rng.integers(2))
    n = int(rng.integers(1, 4))
    ow, oh = (2 * w, 2 * h) if tr else ((w + 1) // 2, (h + 1) // 2)
    d = LayerDesc(IFM_CH=cin, IFM_ROW=w, IFM_COL=h, OFM_CH=cout, OFM_ROW=ow, OFM_COL=oh, SIMD=simd, PE=pe,
                  W_TILES=(cout // pe) * (25 * cin // simd), transposed=tr)
    d.validate()
    env = {}                     # sicn_options fields, per call
    if rng.random() < 0.5:
        env["strip_chunks"] = int(rng.integers(1, 9))
    if rng.random() < 0.5:
        env["tile_x"] = int(rng.choice([16, 32]))
    if rng.random() < 0.3:
        env["split_n"] = int(rng.choice([1, 2, 4]))
    if HAS_ALT and cin == 3 and rng.random() < 0.3:      # the persistent layer-0 kernel (k_l0p, ALT build only), 1 .. all workgroups
        env["l0_form"] = 2
        env.pop("strip_chunks", None)
        env["persistent_grid"] = int(rng.choice([1, 3, 8, 64, 0]))
    if cin != 3 and rng.random() < 0.4:       # the wide persistent kernels (only conv / deconv 128 -> 128 take them; others ignore the request)
        env["wave_tile"] = 128
        env["tile_x"] = 32
        env["persistent_grid"] = int(rng.choice([8, 16, 64, 0]))
    W = rng.integers(-8, 8, (cout, 5, 5, cin)).astype(np.int8)
    b = rng.integers(-128, 128, cout).astype(np.int8)
    words = sicn_ref.pack_finn_tiles(W, simd, pe)
    x = rng.integers(0, 256 if cin == 3 else 128, (n,) + d.in_shape, dtype=np.uint8)
    if cin != 3 and rng.random() < 0.3:
        x.reshape(-1)[::5] |= 0x80
    fpw = api.FixedPointWeights(simd, 4, pe, d.W_TILES, words)
    fn = api.deconv522 if tr else api.conv2d
    got = fn(d, fpw, b, torch.from_numpy(x).cuda(), None, n, options=env or None)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    ref_fn = sicn_ref.deconv522_ref if tr else sicn_ref.conv2d_ref
    ok = all(np.array_equal(got[i], ref_fn(x[i], W, b)) for i in range(n))
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: cin={cin} cout={cout} tr={tr} w={w} h={h} n={n} env={env}", flush=True)
    elif case % 25 == 0:
        print(f"case {case}: ok (cin={cin} cout={cout} tr={tr} {w}x{h} n={n} {env})", flush=True)
print(f"{args.cases - bad}/{args.cases} cases bit-exact")

# whole chains: random image sizes, random nibble weights, latent + reconstruction against the closed form
from simple_image_compression_network_amd.config import eight_layer_descs  # noqa: E402
cbad = 0
for case in range(args.chains):
    env = {}
    if rng.random() < 0.5:
        env["tile_x"] = int(rng.choice([16, 32]))
    if rng.random() < 0.5:
        env["strip_chunks"] = int(rng.integers(1, 6))
    if rng.random() < 0.3:
        env["split_n"] = int(rng.choice([1, 2, 4]))
    if rng.random() < 0.4:
        env["wave_tile"] = 128
        env["tile_x"] = 32
        env["persistent_grid"] = int(rng.choice([8, 16, 64, 0]))
    w, h, n = int(rng.integers(1, 26)) * 16, int(rng.integers(1, 20)) * 16, int(rng.integers(1, 3))
    descs = eight_layer_descs(w, h)
    params_np, params = [], []
    for d in descs:
        Wt = rng.integers(-8, 8, (d.OFM_CH, 5, 5, d.IFM_CH)).astype(np.int8)
        bt = rng.integers(-128, 128, d.OFM_CH).astype(np.int8)
        params_np.append((Wt, bt, d.transposed))
        params.append((api.FixedPointWeights(d.SIMD, 4, d.PE, d.W_TILES, sicn_ref.pack_finn_tiles(Wt, d.SIMD, d.PE)),
                       api.FixedPointWeights(1, 8, 1, d.OFM_CH, bt.view(np.uint8).astype(np.uint64))))
    net = api.EightLayersNet(w, h, params=params, options=env or None)
    x = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    out, lat = net.forward(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    out, lat = out.cpu().numpy(), lat.cpu().numpy()
    ok = True
    for i in range(n):
        ref = sicn_ref.eight_layers_net_ref(x[i], params_np)
        ok = ok and np.array_equal(out[i], ref[7]) and np.array_equal(lat[i], ref[3])
    if not ok:
        cbad += 1
        print(f"CHAIN MISMATCH {case}: {w}x{h} n={n} env={env}", flush=True)
    elif case % 5 == 0:
        print(f"chain {case}: ok ({w}x{h} n={n})", flush=True)
if args.chains:
    print(f"{args.chains - cbad}/{args.chains} chains bit-exact")

# the activation inside the layer kernels (extension beyond the reference): against oracle/sicn_gdn_oracle.c
gbad = 0
if args.gdn:
    from oracle import c_oracle  # noqa: E402  (checker)
for case in range(args.gdn):
    inverse = bool(rng.integers(2))
    beta = rng.integers(1, 65536, 128).astype(np.uint32)
    gamma = rng.integers(0, 128, (128, 128)).astype(np.uint8)
    if rng.random() < 0.5:
        gamma = (gamma >> 4).astype(np.uint8)
    env = {}
    if rng.random() < 0.5:
        env["strip_chunks"] = int(rng.integers(1, 7))
    n = int(rng.integers(1, 4))
    def mk(cin, cout, simd, pe, w, h, tr):
        ow, oh = (2 * w, 2 * h) if tr else ((w + 1) // 2, (h + 1) // 2)
        d = LayerDesc(IFM_CH=cin, IFM_ROW=w, IFM_COL=h, OFM_CH=cout, OFM_ROW=ow, OFM_COL=oh, SIMD=simd, PE=pe,
                      W_TILES=(cout // pe) * (25 * cin // simd), transposed=tr)
        d.validate()
        Wt = rng.integers(-8, 8, (cout, 5, 5, cin)).astype(np.int8)
        bt = rng.integers(-128, 128, cout).astype(np.int8)
        return d, Wt, bt
    if case % 2 == 0 or not HAS_ALT:     # k_l0g (k_l7g below exists in the ALT build only)
        d, Wt, bt = mk(3, 128, 3, 8, int(rng.integers(1, 300)), int(rng.integers(1, 200)), 0)
        x = rng.integers(0, 256, (n,) + d.in_shape, dtype=np.uint8)
        g = api.GDN(beta, gamma, inverse, 12)
        fpw = api.FixedPointWeights(d.SIMD, 4, d.PE, d.W_TILES, sicn_ref.pack_finn_tiles(Wt, d.SIMD, d.PE))
        got = api.conv2d(d, fpw, bt, torch.from_numpy(x).cuda(), None, n, gdn=g, options=env or None).cpu().numpy()
        ok = all(np.array_equal(got[i], c_oracle.gdn(sicn_ref.layer_preact_ref(x[i], Wt, bt, 0), beta, gamma, inverse, 12)) for i in range(n))
        what = f"k_l0g {d.IFM_ROW}x{d.IFM_COL}"
    else:                 # (conv or deconv 128 -> 128, GDN) -> 128 -> RGB with gdn_fuse = 2
        tr = int(rng.integers(2))
        d0, W0, b0 = mk(128, 128, 8, 16, int(rng.integers(1, 70)), int(rng.integers(1, 40)), tr)
        d1, W1, b1 = mk(128, 3, 8, 3, d0.OFM_ROW, d0.OFM_COL, 1)
        params = [(api.FixedPointWeights(d.SIMD, 4, d.PE, d.W_TILES, sicn_ref.pack_finn_tiles(Wt, d.SIMD, d.PE)),
                   api.FixedPointWeights(1, 8, 1, d.OFM_CH, bt.view(np.uint8).astype(np.uint64))) for d, Wt, bt in ((d0, W0, b0), (d1, W1, b1))]
        env["gdn_fuse"] = 2
        net = api.EightLayersNet(descs=[d0, d1], params=params, gdn=[api.GDN(beta, gamma, inverse, 12), None], options=env)
        x = rng.integers(0, 128, (n,) + d0.in_shape, dtype=np.uint8)
        got = net.run_layers(0, 1, torch.from_numpy(x).cuda())[0].cpu().numpy()
        ok = all(np.array_equal(got[i], sicn_ref.deconv522_ref(c_oracle.gdn(sicn_ref.layer_preact_ref(x[i], W0, b0, tr), beta, gamma, inverse, 12), W1, b1))
                 for i in range(n))
        what = f"k_l7g behind {'deconv' if tr else 'conv'} {d0.IFM_ROW}x{d0.IFM_COL}"
    if not ok:
        gbad += 1
        print(f"GDN MISMATCH {case}: {what} n={n} inverse={inverse} env={env}", flush=True)
    elif case % 10 == 0:
        print(f"gdn {case}: ok ({what} n={n} inverse={inverse} {env})", flush=True)
if args.gdn:
    print(f"{args.gdn - gbad}/{args.gdn} fused-activation cases bit-exact")
# the wide kernels' dynamic tile deal: every tile exactly once, whoever takes it
dbad = 0
if args.deal:
    from oracle import c_oracle  # noqa: E402,F811  (checker)
for case in range(args.deal):
    tr = int(rng.integers(2))
    grid = int(rng.choice([8, 16]))
    # position grid (output for the conv, input for the deconv) of tx x ty tiles of 16 x 32, ragged at both edges, with >= 16 tiles per workgroup
    n = int(rng.integers(1, 4))
    need = (16 * grid + n - 1) // n                      # tiles per image
    tx = int(rng.integers(2, 12))
    ty = max(1, (need + tx - 1) // tx + int(rng.integers(0, 3)))
    mw, mh = 32 * (tx - 1) + int(rng.integers(1, 33)), 16 * (ty - 1) + int(rng.integers(1, 17))
    assert ((mw + 31) // 32) * ((mh + 15) // 16) * n >= 16 * grid
    w, h = (mw, mh) if tr else (2 * mw - int(rng.integers(2)), 2 * mh - int(rng.integers(2)))
    ow, oh = (2 * w, 2 * h) if tr else ((w + 1) // 2, (h + 1) // 2)
    d = LayerDesc(IFM_CH=128, IFM_ROW=w, IFM_COL=h, OFM_CH=128, OFM_ROW=ow, OFM_COL=oh, SIMD=8, PE=16, W_TILES=8 * 400, transposed=tr)
    d.validate()
    Wt = rng.integers(-8, 8, (128, 5, 5, 128)).astype(np.int8)
    bt = rng.integers(-128, 128, 128).astype(np.int8)
    fpw = api.FixedPointWeights(8, 4, 16, d.W_TILES, sicn_ref.pack_finn_tiles(Wt, 8, 16))
    net = api.EightLayersNet(descs=[d], params=[(fpw, api.FixedPointWeights(1, 8, 1, 128, bt.view(np.uint8).astype(np.uint64)))],
                             options={"wave_tile": 128, "persistent_grid": grid})
    x = rng.integers(0, 128, (n,) + d.in_shape, dtype=np.uint8)
    xin = torch.from_numpy(x).cuda()
    words = sicn_ref.pack_finn_tiles(Wt, 8, 16)
    ref = np.stack([c_oracle.run_layer(d, words, bt, x[i], form="direct", threads=os.cpu_count() or 1) for i in range(n)])   # the C closed form: these are big
    ok = True
    out = None
    for _ in range(3):
        out, _ = net.run_layers(0, 0, xin)
        torch.cuda.synchronize()
        ok = ok and np.array_equal(out.cpu().numpy(), ref)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            net.run_layers(0, 0, xin, out=out)
    for _ in range(2):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ok = ok and np.array_equal(out.cpu().numpy(), ref)
    if not ok:
        dbad += 1
        print(f"DEAL MISMATCH {case}: tr={tr} {w}x{h} n={n} grid={grid}", flush=True)
    else:
        print(f"deal {case}: ok ({'deconv' if tr else 'conv'} {w}x{h} n={n}, {grid} workgroups)", flush=True)
if args.deal:
    print(f"{args.deal - dbad}/{args.deal} dynamic-deal cases bit-exact")
sys.exit(1 if (bad or cbad or gbad or dbad) else 0)
