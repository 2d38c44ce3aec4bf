#!/usr/bin/env python3
"""k_gdn alone: GDN and IGDN in place over an 8 x 1920 x 1080 x 128 tensor (layer 0's output at 8 x 4K: 2.1 GB), and over the
192-channel latent shape.  usage: gdn_speed.py   (run under rocprofv3 --kernel-trace --stats for per-kernel averages)"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from simple_image_compression_network_amd import api  # noqa: E402
from simple_image_compression_network_amd.hyperprior import random_gdn_params  # noqa: E402

rng = np.random.default_rng(0)
for c, npos in ((128, 8 * 1920 * 1080), (192, 8 * 240 * 135)):
    x = torch.randint(0, 256, (npos, c), dtype=torch.uint8, device="cuda")
    for inverse in (False, True):
        beta, gamma = random_gdn_params(rng, c)
        g = api.GDN(beta, gamma, inverse=inverse, shift=12)
        for _ in range(2):
            g.apply_(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.apply_(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"C={c} {'IGDN' if inverse else 'GDN '} {npos} positions: {dt * 1e3:.3f} ms  {2 * npos * c / dt / 1e9:.0f} GB/s (r+w)  "
              f"{npos * c / dt / 1e9:.1f} Gelement/s", flush=True)
