#!/usr/bin/env python3
"""In-process interleaved A/B of sicn_options sets on the 8 x 4K batch (or --width/--height/--images): per-layer device ms
(sicn_net_layer_ms) and whole-forward ms, outputs compared between the sets.
usage: ab_options.py "wave_tile=64" "wave_tile=128" ...   ("" = the defaults)"""
import argparse
import statistics
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simple_image_compression_network_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("sets", nargs="+")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--images", type=int, default=8)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--no-check", action="store_true", help="experiment builds that give wrong results on purpose")
args = ap.parse_args()

W, H, B = args.width, args.height, args.images
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
nets = []
for s in args.sets:
    opts = {k: int(v) for k, v in (a.split("=") for a in s.split())} if s.strip() else None
    nets.append(api.EightLayersNet(W, H, options=opts))
out = torch.empty((B,) + nets[0].descs[-1].out_shape, dtype=torch.uint8, device="cuda")
lat = torch.empty((B,) + nets[0].descs[3].out_shape, dtype=torch.uint8, device="cuda")
ref = None
for s, net in zip(args.sets, nets):
    net.forward(x, out, lat)
    torch.cuda.synchronize()
    if ref is None:
        ref = (out.clone(), lat.clone())
    elif not args.no_check:
        assert torch.equal(out, ref[0]) and torch.equal(lat, ref[1]), f"'{s}' changes the result"
    net.profile(True)
    net.layer_ms(reset=True)
wall = {s: [] for s in args.sets}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(args.rounds):
    for s, net in zip(args.sets, nets):
        e0.record()
        for _ in range(args.reps):
            net.forward(x, out, lat)
        e1.record()
        torch.cuda.synchronize()
        wall[s].append(e0.elapsed_time(e1) / args.reps)
for s, net in zip(args.sets, nets):
    ms, cnt = net.layer_ms()
    per = " ".join(f"L{i}={m / max(c, 1):.3f}" for i, (m, c) in enumerate(zip(ms, cnt)))
    print(f"[{s or 'defaults'}] forward median {statistics.median(wall[s]):.3f} ms  min {min(wall[s]):.3f}   {per}", flush=True)
