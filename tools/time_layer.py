#!/usr/bin/env python3
"""Time ONE layer of the 4K net standalone (NHWC in/out through sicn_conv2d / sicn_deconv522), to compare
with its in-chain time (internal layouts).  usage: time_layer.py LAYER [--images 8]"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simple_image_compression_network_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("layer", type=int)
ap.add_argument("--images", type=int, default=8)
args = ap.parse_args()
net = api.EightLayersNet(3840, 2160)
l = args.layer
d = net.descs[l]
x = torch.randint(0, 128, (args.images,) + d.in_shape, dtype=torch.uint8, device="cuda")
out, _ = net.run_layers(l, l, x)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
best = 1e9
for r in range(6):
    ev[0].record()
    for _ in range(3):
        net.run_layers(l, l, x)
    ev[1].record()
    torch.cuda.synchronize()
    best = min(best, ev[0].elapsed_time(ev[1]) / 3)
print(f"layer {l} standalone NHWC: {best:.3f} ms")
