#!/usr/bin/env python3
"""Probe: how do the HBM-bound layers (L0, L7) and the MFMA-bound ones (L1..L6) scale with the number of CUs a stream may use
(hipExtStreamCreateWithCUMask), and what does running them side by side on complementary masks give?
usage: cumask_probe.py [n_images=8]"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from simple_image_compression_network_amd import api  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H = 3840, 2160


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return s.value


netA = api.EightLayersNet(W, H)
netB = api.EightLayersNet(W, H)
d = netA.descs
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)).cuda()
a0 = torch.empty((n,) + d[0].out_shape, dtype=torch.uint8, device="cuda")
a6 = torch.empty((n,) + d[6].out_shape, dtype=torch.uint8, device="cuda")
out = torch.empty((n,) + d[7].out_shape, dtype=torch.uint8, device="cuda")
netA.run_layers(0, 0, x, out=a0)
netA.run_layers(1, 6, a0, out=a6)
torch.cuda.synchronize()


def timed(fns_streams, reps=10):
    """fns_streams: [(fn(stream), stream)]: each enqueues `reps` calls on its own stream; returns wall ms for all to finish."""
    for fn, s in fns_streams:
        fn(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for fn, s in fns_streams:
            fn(s)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


hbm = lambda s: (netB.run_layers(0, 0, x, out=a0, stream=s), netB.run_layers(7, 7, a6, out=out, stream=s))
l0 = lambda s: netB.run_layers(0, 0, x, out=a0, stream=s)
l7 = lambda s: netB.run_layers(7, 7, a6, out=out, stream=s)
mfma = lambda s: netA.run_layers(1, 6, a0, out=a6, stream=s)

full = masked_stream(range(256))
print(f"{n} x 4K   full mask: L0 {timed([(l0, full)]):.3f}  L7 {timed([(l7, full)]):.3f}  L1-6 {timed([(mfma, full)]):.3f} ms", flush=True)
print(f"serial (one stream, L0+L7 then L1-6): {timed([(hbm, full), (mfma, full)]):.3f} ms", flush=True)
for kind in ("strided", "first"):
    for k in (32, 64, 96, 128):
        if kind == "strided":
            step = 256 // k
            bits_b = [i for i in range(256) if i % step == 0]
        else:
            bits_b = list(range(k))
        bits_a = [i for i in range(256) if i not in set(bits_b)]
        sb, sa = masked_stream(bits_b), masked_stream(bits_a)
        t_l0, t_l7 = timed([(l0, sb)]), timed([(l7, sb)])
        t_m = timed([(mfma, sa)])
        t_both = timed([(hbm, sb), (mfma, sa)])
        print(f"{kind:8s} {k:3d} CUs for L0/L7: L0 {t_l0:.3f}  L7 {t_l7:.3f} | L1-6 on {256 - k}: {t_m:.3f} | side by side: {t_both:.3f} ms",
              flush=True)
# no masks, two ordinary streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print(f"two plain streams side by side: {timed([(hbm, s1.cuda_stream), (mfma, s2.cuda_stream)]):.3f} ms", flush=True)
