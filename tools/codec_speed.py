#!/usr/bin/env python3
"""Throughput of the latent container / rANS coder (extension beyond the reference) on a 4K latent."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simple_image_compression_network_amd import api, codec  # noqa: E402

W, H = 3840, 2160
net = api.EightLayersNet(W, H)
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1, H, W, 3), dtype=np.uint8)).cuda()
_, lat = net.forward(x)
torch.cuda.synchronize()
lat0 = lat[0].contiguous()
n = lat0.numel()
for mode, name in ((codec.RAW8, "raw8"), (codec.PACKED7, "packed7"), (codec.RANS, "rANS"), (codec.RANSW, "rANS-W")):
    c = codec.encode_latent(lat0, W, H, mode)
    back, _ = codec.decode_latent(c)
    torch.cuda.synchronize()
    assert torch.equal(back, lat0)
    t0 = time.perf_counter()
    for _ in range(20):
        c = codec.encode_latent(lat0, W, H, mode)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        codec.decode_latent(c)
    torch.cuda.synchronize()
    td = (time.perf_counter() - t0) / 20
    print(f"{name:8s}: {n} symbols -> {c.numel()} bytes ({8 * c.numel() / n:.2f} bit/symbol); "
          f"encode {te * 1e3:.3f} ms ({n / te / 1e9:.2f} Gsym/s), decode {td * 1e3:.3f} ms ({n / td / 1e9:.2f} Gsym/s)  [host-synchronous calls]")

# batches: 8 latents per call
lat8 = lat0[None].repeat(8, 1, 1, 1).contiguous()
slots, sizes = codec.encode_latents(lat8, W, H)
back, _ = codec.decode_latents(slots, sizes)
torch.cuda.synchronize()
assert torch.equal(back, lat8)
t0 = time.perf_counter()
for _ in range(10):
    slots, sizes = codec.encode_latents(lat8, W, H)
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / 10
t0 = time.perf_counter()
for _ in range(10):
    codec.decode_latents(slots, sizes)
torch.cuda.synchronize()
td = (time.perf_counter() - t0) / 10
print(f"rANS-W batch of 8: encode {te * 1e3:.3f} ms ({8 * n / te / 1e9:.2f} Gsym/s), decode {td * 1e3:.3f} ms ({8 * n / td / 1e9:.2f} Gsym/s)")
