#!/usr/bin/env python3
"""Speed of the generic ConvLayer_Batch surface on a few shapes."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simple_image_compression_network_amd import api, convlayer as cl  # noqa: E402

rng = np.random.default_rng(0)
for (K, C, D, O, reps) in [(3, 64, 130, 64, 4), (5, 128, 68, 128, 4), (1, 256, 64, 256, 4), (3, 16, 258, 32, 4)]:
    simd, pe = min(C, 8), min(O, 8)
    desc = cl.ConvLayerDesc(K=K, IFM_CH=C, IFM_DIM=D, OFM_CH=O, SIMD=simd, PE=pe, W_BIT=4, IN_SIGNED=False, OUT_BIT=32)
    w = rng.integers(-8, 8, (O, K * K * C)).astype(np.int64)
    nf, sf = O // pe, K * K * C // simd
    el = (w.reshape(nf, pe, sf, simd) & 15).astype(np.uint64)
    words = (el << (np.arange(simd, dtype=np.uint64) * np.uint64(4))[None, None, None, :]).sum(axis=3, dtype=np.uint64)
    words = np.ascontiguousarray(words.transpose(1, 0, 2).reshape(pe, nf * sf))
    fpw = api.FixedPointWeights(simd, 4, pe, desc.W_TILES, words)
    act = cl.PassThroughActivation(ACC_BIT=32, ACC_SIGNED=True)
    x = torch.from_numpy(rng.integers(0, 256, (reps, D, D, C), dtype=np.uint8)).cuda()
    out = cl.ConvLayer_Batch(desc, x, None, fpw, act, reps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        cl.ConvLayer_Batch(desc, x, out, fpw, act, reps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    macs = reps * desc.OFM_DIM ** 2 * O * K * K * C
    # the same with the parameters resident (cl.ConvLayer): what the kernel itself takes (events around 20 launches)
    layer = cl.ConvLayer(desc, fpw, act)
    layer(x, out, reps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        layer(x, out, reps)
    e1.record()
    torch.cuda.synchronize()
    dk = e0.elapsed_time(e1) / 20 * 1e-3
    layer.close()
    print(f"K={K} C={C} D={D} O={O} reps={reps}: per call with upload {dt * 1e3:.3f} ms ({2 * macs / dt / 1e12:.2f} TOP/s); parameters resident "
          f"{dk * 1e6:.1f} us ({2 * macs / dk / 1e12:.1f} TOP/s)")
