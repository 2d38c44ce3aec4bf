#!/usr/bin/env python3
"""One configuration in a loop, for `rocprofv3 --kernel-trace --stats` (per-kernel averages of exactly that configuration).
usage: trace_config.py WIDTH HEIGHT N_IMAGES [coded] [iters=100] [sicn_options k=v ...]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from simple_image_compression_network_amd import api, codec  # noqa: E402

w, h, n = (int(a) for a in sys.argv[1:4])
rest = sys.argv[4:]
coded = "coded" in rest
kv = dict(a.split("=") for a in rest if "=" in a)
iters = int(kv.pop("iters", 100))
opts = {k: int(v) for k, v in kv.items()}
net = api.EightLayersNet(w, h, options=opts or None)
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
out = torch.empty((n,) + net.descs[-1].out_shape, dtype=torch.uint8, device="cuda")
lat = torch.empty((n,) + net.descs[3].out_shape, dtype=torch.uint8, device="cuda")
lat2 = torch.empty_like(lat)
coder = codec.LatentCoder(n, *net.descs[3].out_shape, image_width=w, image_height=h) if coded else None
for _ in range(iters):
    if coded:
        net.analysis(x, lat)
        coder.encode(lat)
        coder.decode(lat2)
        net.synthesis(lat2, out)
    else:
        net.forward(x, out, lat)
torch.cuda.synchronize()
if coded:
    coder.check()
    print("bytes per image:", coder.sizes()[:4])
print("done", w, h, n, "coded" if coded else "", opts)
