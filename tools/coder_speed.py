#!/usr/bin/env python3
"""Device time of the rANS-W coder alone (encode, decode: asynchronous batch calls in a loop) on a latent produced by the
analysis half.  usage: coder_speed.py WIDTH HEIGHT N_IMAGES"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from simple_image_compression_network_amd import api, codec  # noqa: E402

w, h, n = (int(a) for a in sys.argv[1:4])
net = api.EightLayersNet(w, h)
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
lat = torch.empty((n,) + net.descs[3].out_shape, dtype=torch.uint8, device="cuda")
lat2 = torch.empty_like(lat)
net.analysis(x, lat)
coder = codec.LatentCoder(n, *net.descs[3].out_shape, image_width=w, image_height=h)


def t_of(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


te = t_of(lambda: coder.encode(lat))
td = t_of(lambda: coder.decode(lat2))
coder.check()
assert torch.equal(lat, lat2)
print(f"{w}x{h} x{n}: encode {te:.1f} us  decode {td:.1f} us  bytes/image {coder.sizes()[0]}", flush=True)
