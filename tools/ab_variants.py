#!/usr/bin/env python3
"""In-process interleaved A/B of kernel variants (one device, one process: cdna guide rule 24).
usage: ab_variants.py ENVVAR v0 v1 [...] [--rounds R] — per-layer median ms for each value."""
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
for v in args.values:
    os.environ[args.envvar] = v
    net.forward(x, out, lat)
torch.cuda.synchronize()
net.profile(True)
res = {v: [] for v in args.values}
for r in range(args.rounds):
    for v in args.values:
        os.environ[args.envvar] = v
        net.layer_ms(reset=True)
        for _ in range(3):
            net.forward(x, out, lat)
        ms, cnt = net.layer_ms(reset=True)
        res[v].append([m / c for m, c in zip(ms, cnt)])
for v in args.values:
    med = [statistics.median(r[l] for r in res[v]) for l in range(8)]
    print(f"{args.envvar}={v}: total {sum(med):.3f} ms | " + " ".join(f"L{l}={m:.3f}" for l, m in enumerate(med)))
