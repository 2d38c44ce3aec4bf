#!/usr/bin/env python3
"""Diagnostic (SICN_LIB = a build with -DSICN_EXP_L7_STAMP): cycles per phase of a k_l7 step, averaged over all waves and
steps of the 8 x 4K forward pass: row requests | fragment reads + MFMAs | pack + stores | wait for the next rows | barrier."""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simple_image_compression_network_amd import _lib, api  # noqa: E402

W, H, B = 3840, 2160, 8
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
net = api.EightLayersNet(W, H)
out = torch.empty((B,) + net.descs[-1].out_shape, dtype=torch.uint8, device="cuda")
L = ctypes.CDLL(_lib.lib()._name)
buf = (ctypes.c_ulonglong * 8)()
for _ in range(3):
    net.forward(x, out, want_latent=False)
torch.cuda.synchronize()
L.sicn_debug_l7_stamps(buf)
reps = 10
for _ in range(reps):
    net.forward(x, out, want_latent=False)
torch.cuda.synchronize()
L.sicn_debug_l7_stamps(buf)
steps = buf[5]                      # wave-steps summed over lane-0 of every wave
names = ["requests", "reads+mfma", "pack+stores", "wait rows", "barrier"]
tot = sum(buf[i] for i in range(5))
print(f"wave-steps {steps}, cycles per step {tot / steps:.0f} (100 MHz ticks x ... s_memtime units)")
for i, n in enumerate(names):
    print(f"  {n:12s} {buf[i] / steps:8.1f}")
