#!/usr/bin/env python3
"""In-process interleaved A/B of settings that change the SCHEDULE (not one kernel): wall time of whole forward
passes, measured with events on the caller's stream.  usage: ab_wall.py ENVVAR v0 v1 [...] [--rounds R] [--images N]"""
import argparse
import os
import statistics
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simple_image_compression_network_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("envvar")
ap.add_argument("values", nargs="+")
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--images", type=int, default=8)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
args = ap.parse_args()

W, H, B = args.width, args.height, args.images
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
net = api.EightLayersNet(W, H)
out = torch.empty((B,) + net.descs[-1].out_shape, dtype=torch.uint8, device="cuda")
lat = torch.empty((B,) + net.descs[3].out_shape, dtype=torch.uint8, device="cuda")
ref = None
for v in args.values:
    os.environ[args.envvar] = v
    net.forward(x, out, lat)
    torch.cuda.synchronize()
    if ref is None:
        ref = (out.clone(), lat.clone())
    else:
        assert torch.equal(out, ref[0]) and torch.equal(lat, ref[1]), f"{args.envvar}={v} changes the result"
res = {v: [] for v in args.values}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(args.rounds):
    for v in args.values:
        os.environ[args.envvar] = v
        e0.record()
        for _ in range(4):
            net.forward(x, out, lat)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 4)
for v in args.values:
    print(f"{args.envvar}={v}: median {statistics.median(res[v]):.3f} ms/forward  min {min(res[v]):.3f}  ({B} images {W}x{H})")
