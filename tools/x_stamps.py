#!/usr/bin/env python3
"""Cycles per tile, hand-over share and the clock the chip holds for the wide persistent kernels (k_mfma16x.hip) of the 8 x 4K
net, every layer measured INSIDE the whole forward pass (its real input / output layouts).  Diagnostic build (-DSICN_STAMP,
gpurun_build/libsicn_stamp.so); in the product build no stamp exists.  The deconv's hand-over figure is the one of four per
tile that belongs to the tile's last phase."""
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
out = torch.empty((n,) + net.descs[-1].out_shape, dtype=torch.uint8, device="cuda")
lat = torch.empty((n,) + net.descs[3].out_shape, dtype=torch.uint8, device="cuda")
buf = torch.zeros((4 * 2048,), dtype=torch.int64, device="cuda")
for _ in range(10):                       # warm: the chip settles at the clock it holds under this load
    net.forward(x, out, lat)
torch.cuda.synchronize()
assert L.sicn_debug_stamp_buffer_x(ctypes.c_void_p(buf.data_ptr())) == 0
net.profile(True)
net.layer_ms(reset=True)
net.forward(x, out, lat)
torch.cuda.synchronize()
ms, cnt = net.layer_ms()
allr = buf.cpu().numpy().reshape(4, 256, 8).astype(np.float64)
for slot, layer in enumerate((1, 2, 6, 5)):
    s = allr[slot]
    s = s[s[:, 0] > 0]
    if not len(s):
        continue
    cyc, rt, tiles, hand = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    clk = cyc / rt * 0.1
    print(f"layer {layer}: {ms[layer] / max(cnt[layer], 1):.3f} ms; {len(s)} workgroups x {tiles.mean():.1f} tiles; "
          f"{(cyc / tiles).mean():.0f} cycles per tile, hand-over pass {(hand / tiles).mean():.0f}; clock {np.median(clk):.2f} GHz; "
          f"spans p5 {np.percentile(cyc, 5):.0f} p50 {np.percentile(cyc, 50):.0f} max {cyc.max():.0f}")

# ---- round 5: where layer 5's "8 % per tile" against layer 6 (and layer 2's against layer 1) comes from -------------------------------------
# The same kernel walks 8 tiles per workgroup in layers 2 / 5 and 32 in layers 1 / 6.  Model: launch-to-end time = a + b * (tiles per workgroup);
# b from the pair of layers, a = what a launch costs beyond its tiles (launch latency, the first tile's un-overlapped patch + ring fill, the last
# hand-over and the drain).  The stamps give the same split from inside: a workgroup's own span against tiles * (cycles per tile of the long layer),
# and the start-to-end window of all workgroups against the layer's event time.
def fit(short, long_):
    (ms_s, t_s), (ms_l, t_l) = short, long_
    b = (ms_l - ms_s) / (t_l - t_s)
    return b, ms_s - b * t_s
per = {}
for slot, layer in enumerate((1, 2, 6, 5)):
    s = allr[slot]
    s = s[s[:, 0] > 0]
    if len(s):
        per[layer] = (ms[layer] / max(cnt[layer], 1), s[:, 2].mean(), (s[:, 0] / s[:, 2]).mean(), np.median(s[:, 0] / s[:, 1] * 0.1))   # (s_memtime start stamps of different CUs are not comparable: no window)
for short, long_, name in ((2, 1, "conv"), (5, 6, "deconv")):
    if short in per and long_ in per:
        b, a = fit(per[short][:2], per[long_][:2])
        print(f"{name}: layer {long_} {per[long_][0]:.3f} ms / {per[long_][1]:.1f} tiles per workgroup, layer {short} {per[short][0]:.3f} ms / {per[short][1]:.1f}:  "
              f"time = {a * 1e3:.1f} us + {b * 1e3:.2f} us per tile;  the fixed part is {100 * a / per[short][0]:.1f} % of layer {short} and "
              f"{100 * a / per[long_][0]:.1f} % of layer {long_}")
        for l in (long_, short):
            inside = per[l][1] * per[l][2] / per[l][3] / 1e3   # us a workgroup spends on its tiles at the clock it saw
            print(f"   layer {l}: {per[l][2]:.0f} cycles per tile inside the workgroups at {per[l][3]:.2f} GHz = {inside:.1f} us of the event's {per[l][0] * 1e3:.1f} us "
                  f"({per[l][0] * 1e3 - inside:.1f} us outside a workgroup's own span: launch, start skew, the slowest workgroup)")

# ---- per-XCD balance of the static tile deal: does one XCD finish later than the others? (spans in cycles and in 100 MHz ticks = real time) ----
for slot, layer in enumerate((1, 2, 6, 5)):
    s = allr[slot]
    s = s[s[:, 0] > 0]
    if not len(s):
        continue
    xs = np.unique(s[:, 6])
    rows = []
    for xcc in xs:
        t = s[s[:, 6] == xcc]
        rows.append((int(xcc) & 15, len(t), t[:, 2].mean(), t[:, 1].mean() / 100.0, t[:, 1].max() / 100.0, np.median(t[:, 0] / t[:, 1] * 0.1)))
    worst = max(r[4] for r in rows)
    print(f"layer {layer}: per XCD (id: workgroups, tiles each, mean span us, max span us, GHz): " +
          "; ".join(f"{r[0]}: {r[1]}, {r[2]:.1f}, {r[3]:.0f}, {r[4]:.0f}, {r[5]:.2f}" for r in rows) +
          f"  | all: mean {s[:, 1].mean() / 100.0:.0f} us, max {worst:.0f} us ({100 * (worst / (s[:, 1].mean() / 100.0) - 1):.1f} % above the mean)")
