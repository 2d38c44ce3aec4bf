#!/usr/bin/env python3
"""Stress for the wide kernels' dynamic tile deal (k_mfma16x.hip DealX): 8 x 4K forward passes with the deal (default) against the round-2
kernels (wave_tile = 64: no persistent workgroups at all) on random inputs, while a THIRD net keeps another stream busy so that the XCDs'
progress — and with it who takes which ticket and who steals — differs from pass to pass.  Any difference in latent or reconstruction is a
bug (a lost or doubled tile).  usage: stress_deal.py [iterations]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simple_image_compression_network_amd import api  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, w, h = 8, 3840, 2160
rng = np.random.default_rng(7)
a = api.EightLayersNet(w, h)
b = api.EightLayersNet(w, h, options={"wave_tile": 64})
c = api.EightLayersNet(w, h)              # the background load: its own workspace, its own stream
side = torch.cuda.Stream()
xc = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
oc = torch.empty((n,) + c.descs[-1].out_shape, dtype=torch.uint8, device="cuda")
x = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
bad = 0
for it in range(iters):
    x.copy_(torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)))
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(1 + it % 3):
            c.forward(xc, oc, want_latent=False, stream=side)
    oa, la = a.forward(x)
    torch.cuda.synchronize()
    ob, lb = b.forward(x)
    torch.cuda.synchronize()
    if not (torch.equal(oa, ob) and torch.equal(la, lb)):
        bad += 1
        print(f"MISMATCH iter {it}: out {int((oa != ob).sum())} latent {int((la != lb).sum())} bytes", flush=True)
    if it % 5 == 4:
        print(f"{it + 1} iterations, mismatches so far {bad}", flush=True)
print("stress_deal:", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
