#!/usr/bin/env python3
"""L0 / L7 device time against sicn_options.strip_chunks at one image size.  usage: strip_sweep.py WIDTH HEIGHT N_IMAGES"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from simple_image_compression_network_amd import api  # noqa: E402

w, h, n = (int(a) for a in sys.argv[1:4])
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
for chunks in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 135):
    net = api.EightLayersNet(w, h, options={"strip_chunks": chunks} if chunks else None)
    out = torch.empty((n,) + net.descs[-1].out_shape, dtype=torch.uint8, device="cuda")
    lat = torch.empty((n,) + net.descs[3].out_shape, dtype=torch.uint8, device="cuda")
    net.profile(True)
    for _ in range(5):
        net.forward(x, out, lat)
    net.layer_ms(reset=True)
    for _ in range(30):
        net.forward(x, out, lat)
    ms, cnt = net.layer_ms()
    print(f"chunks {chunks:4d}: L0 {1e3 * ms[0] / cnt[0]:7.1f} us   L7 {1e3 * ms[7] / cnt[7]:7.1f} us", flush=True)
