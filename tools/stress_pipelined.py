#!/usr/bin/env python3
"""Stress: the software-pipelined kernels (k_mfma16p.hip, default) against the plain ones (prefetch = 1) on many random
inputs, full-size and small grids, whole chains — any difference is a bug (both are bit-exact vs the oracle in the tests;
this hunts for timing-dependent ones).  usage: stress_pipelined.py [iterations]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simple_image_compression_network_amd import api  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(123)
bad = 0
for (w, h, n) in [(3840, 2160, 2), (1920, 1080, 1), (768, 512, 3), (1024, 1024, 2), (208, 112, 5)]:
    a = api.EightLayersNet(w, h)
    b = api.EightLayersNet(w, h, options={"prefetch": 1})
    for it in range(iters):
        x = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
        oa, la = a.forward(x)
        ob, lb = b.forward(x)
        torch.cuda.synchronize()
        if not (torch.equal(oa, ob) and torch.equal(la, lb)):
            bad += 1
            print(f"MISMATCH {w}x{h} n={n} iter {it}: out {int((oa != ob).sum())} latent {int((la != lb).sum())} bytes", flush=True)
    print(f"{w}x{h} x{n}: {iters} iterations done, mismatches so far {bad}", flush=True)
print("stress:", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
