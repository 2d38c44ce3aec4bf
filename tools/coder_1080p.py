#!/usr/bin/env python3
"""One 1080p image through analysis -> rANS-W encode -> decode -> synthesis, eagerly, N times: run under
`rocprofv3 --kernel-trace --stats` to see which coder kernels the small-image latency is made of."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simple_image_compression_network_amd import api, codec  # noqa: E402

W, H, N = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 20
SS = (None if len(sys.argv) < 3 else sys.argv[2] if sys.argv[2] == "auto" else int(sys.argv[2]))
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1, H, W, 3), dtype=np.uint8)).cuda()
net = api.EightLayersNet(W, H)
lat = torch.empty((1,) + net.descs[3].out_shape, dtype=torch.uint8, device="cuda")
lat2 = torch.empty_like(lat)
out = torch.empty((1,) + net.descs[-1].out_shape, dtype=torch.uint8, device="cuda")
coder = codec.LatentCoder(1, *net.descs[3].out_shape, image_width=W, image_height=H, device="cuda", stream_symbols=SS)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(2):
    e0.record()
    for _ in range(N):
        net.analysis(x, lat)
        coder.encode(lat)
        coder.decode(lat2)
        net.synthesis(lat2, out)
    e1.record()
    torch.cuda.synchronize()
coder.check()
assert torch.equal(lat, lat2)
print(f"{e0.elapsed_time(e1) / N:.4f} ms per image, {sum(coder.sizes())} bytes")
