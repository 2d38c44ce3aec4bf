#!/usr/bin/env python3
"""GPU-only speed of the small BASELINE configs (1: L0 768x512, 2: 256^2 analysis, 3: 1080p enc+dec, with and without
the rANS-W coder), eager and as one hipGraph, plus per-layer device times.  usage: config_speed.py [sicn_options k=v ...]"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from simple_image_compression_network_amd import api, codec  # noqa: E402

opts = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:])}


def t_of(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


res = {"options": opts}
for name, (w, h, n) in {"768x512": (768, 512, 1), "256x256": (256, 256, 1), "1080p": (1920, 1080, 1), "1080p_x4": (1920, 1080, 4)}.items():
    net = api.EightLayersNet(w, h, options=opts or None)
    x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
    out = torch.empty((n,) + net.descs[-1].out_shape, dtype=torch.uint8, device="cuda")
    lat = torch.empty((n,) + net.descs[3].out_shape, dtype=torch.uint8, device="cuda")
    lat2 = torch.empty_like(lat)
    coder = codec.LatentCoder(n, *net.descs[3].out_shape, image_width=w, image_height=h)
    t_eager = t_of(lambda: net.forward(x, out, lat))
    g = net.capture(x, out, lat)
    t_graph = t_of(g.replay)

    def coded():
        net.analysis(x, lat)
        coder.encode(lat)
        coder.decode(lat2)
        net.synthesis(lat2, out)
    t_coded = t_of(coded)
    gc2 = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gc2, stream=side):
            coded()
    t_coded_graph = t_of(gc2.replay)
    coder.check()
    net.profile(True)
    net.layer_ms(reset=True)
    for _ in range(20):
        net.forward(x, out, lat)
    ms, cnt = net.layer_ms()
    net.profile(False)
    px = n * w * h
    ops = sum(2.0 * d.algorithmic_macs for d in net.descs) * n
    res[name] = {"eager_ms": round(t_eager * 1e3, 4), "graph_ms": round(t_graph * 1e3, 4), "coded_ms": round(t_coded * 1e3, 4),
                 "coded_graph_ms": round(t_coded_graph * 1e3, 4),
                 "Mpx_s_eager": round(px / t_eager / 1e6), "Mpx_s_graph": round(px / t_graph / 1e6),
                 "Mpx_s_coded": round(px / t_coded / 1e6), "Mpx_s_coded_graph": round(px / t_coded_graph / 1e6),
                 "mfma_frac_graph": round(ops / t_graph / 5e15, 4),
                 "layer_us": [round(1e3 * m / max(c, 1), 1) for m, c in zip(ms, cnt)]}
print(json.dumps(res, indent=1))
