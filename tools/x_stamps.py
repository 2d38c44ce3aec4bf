#!/usr/bin/env python3
"""Cycles per tile, hand-over share and the clock the chip holds for the wide persistent kernels (k_mfma16x.hip) on the 8 x 4K
batch.  Diagnostic build (-DSICN_STAMP, gpurun_build/libsicn_stamp.so); in the product build no stamp exists.
usage: x_stamps.py [layer ...]   (default: 1 6)"""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("SICN_LIB", str(ROOT / "gpurun_build" / "libsicn_stamp.so"))
from simple_image_compression_network_amd import _lib, api  # noqa: E402

L = _lib.lib()
n, W, H = 8, 3840, 2160
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, H, W, 3), dtype=np.uint8)).cuda()
L.sicn_debug_stamp_buffer_x.argtypes = [ctypes.c_void_p]
net = api.EightLayersNet(W, H)
layers = [int(a) for a in sys.argv[1:]] or [1, 6]
for layer in layers:
    outl = torch.empty((n,) + net.descs[layer].out_shape, dtype=torch.uint8, device="cuda")
    buf = torch.zeros((256 * 8,), dtype=torch.int64, device="cuda")
    null = torch.zeros((256 * 8,), dtype=torch.int64, device="cuda")
    for _ in range(10):                       # warm: the chip settles at the clock it holds under this load
        net.run_layers(0, layer, x, out=outl)
    torch.cuda.synchronize()
    # stamp only the layer of interest: every wide launch of the chain writes the buffer, the last one wins — run the chain up to it
    assert L.sicn_debug_stamp_buffer_x(ctypes.c_void_p(buf.data_ptr())) == 0
    net.profile(True)
    net.layer_ms(reset=True)
    net.run_layers(0, layer, x, out=outl)
    torch.cuda.synchronize()
    ms, cnt = net.layer_ms()
    s = buf.cpu().numpy().reshape(256, 8).astype(np.float64)
    s = s[s[:, 0] > 0]
    cyc, rt, tiles, hand = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    clk = cyc / rt * 0.1
    start, end = s[:, 4], s[:, 4] + cyc
    print(f"layer {layer}: {ms[layer] / max(cnt[layer], 1):.3f} ms; {len(s)} workgroups x {tiles.mean():.1f} tiles; "
          f"{(cyc / tiles).mean():.0f} cycles per tile, hand-over {(hand / np.maximum(tiles - 1, 1)).mean():.0f}; clock {np.median(clk):.2f} GHz; "
          f"spans p5 {np.percentile(cyc, 5):.0f} p50 {np.percentile(cyc, 50):.0f} max {cyc.max():.0f}; "
          f"first start .. last end {end.max() - start.min():.0f} cycles")
