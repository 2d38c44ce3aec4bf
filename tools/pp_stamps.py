#!/usr/bin/env python3
"""Cycles per tile and the clock the chip holds: layer 1 (8 x 4K) by the one-tile-per-workgroup kernel (prefetch = 2) and by the
persistent one (prefetch = 3).  Diagnostic build, see tools/pass_stamps.py.  s_memtime counts shader cycles, s_memrealtime 100 MHz."""
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
L.sicn_debug_stamp_buffer.argtypes = [ctypes.c_void_p]
for mode in (2, 3):
    net = api.EightLayersNet(W, H, options={"prefetch": mode})
    out1 = torch.empty((n,) + net.descs[1].out_shape, dtype=torch.uint8, device="cuda")
    slots = 16320 + 64
    buf = torch.zeros((slots * 4 * 8,), dtype=torch.int64, device="cuda")
    assert L.sicn_debug_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    for _ in range(20):                       # warm: the chip settles at the clock it holds under this load
        net.run_layers(0, 1, x, out=out1)     # layer 0 -> (GROUP layout, as inside the net) -> layer 1
    torch.cuda.synchronize()
    buf.zero_()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    net.profile(True)
    net.layer_ms(reset=True)
    t0.record()
    net.run_layers(0, 1, x, out=out1)
    t1.record()
    torch.cuda.synchronize()
    ms, cnt = net.layer_ms()
    print(f"layer 1 in the chain: {ms[1] / max(cnt[1], 1):.3f} ms")
    s = buf.cpu().numpy().reshape(slots, 4, 8)
    s = s[s[:, 0, 0] > 0][:, 0, :].astype(np.float64)
    if mode == 2:
        life, loop, loop_rt = s[:, 0], s[:, 2], s[:, 3]
        clk = loop / loop_rt * 0.1
        print(f"one tile per workgroup: {t0.elapsed_time(t1):.3f} ms; life {life.mean():.0f} cycles per tile, loop {loop.mean():.0f}; "
              f"clock inside the loop {np.median(clk):.2f} GHz")
    else:
        cyc, rt, tiles = s[:, 0], s[:, 1], s[:, 2]
        clk = cyc / rt * 0.1
        start, end = s[:, 5], s[:, 5] + cyc
        print(f"  workgroup spans: duration p5 {np.percentile(cyc, 5):.0f} p50 {np.percentile(cyc, 50):.0f} p95 {np.percentile(cyc, 95):.0f} max {cyc.max():.0f} cycles; "
              f"starts spread over {np.percentile(start, 99) - np.percentile(start, 1):.0f} cycles; "
              f"p1 start .. p99 end {np.percentile(end, 99) - np.percentile(start, 1):.0f} cycles")
        np.savez_compressed(ROOT / "gpurun_out" / "pp_stamps.npz", s=s)
        per_tile = cyc / tiles
        print(f"  cycles per tile by workgroup: p5 {np.percentile(per_tile, 5):.0f} p50 {np.percentile(per_tile, 50):.0f} p95 {np.percentile(per_tile, 95):.0f} max {per_tile.max():.0f}")
        print(f"persistent: {t0.elapsed_time(t1):.3f} ms; {len(s)} workgroups x {tiles.mean():.1f} tiles, {(cyc / tiles).mean():.0f} cycles per tile "
              f"({(cyc / tiles / 50).mean():.0f} per pass); clock {np.median(clk):.2f} GHz")
