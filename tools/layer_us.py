#!/usr/bin/env python3
"""Per-layer device time (us) and whole-forward time (hipGraph replay) of the small configs, for one library build
(SICN_LIB) and several option sets.  usage: layer_us.py "k=v k=v" "" ...      ("" = the defaults)"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from simple_image_compression_network_amd import api  # noqa: E402

sets = sys.argv[1:] or [""]
CONFIGS = {"256x256": (256, 256, 1), "768x512": (768, 512, 1), "1080p": (1920, 1080, 1)}
ref = {}
for s in sets:
    opts = {k: int(v) for k, v in (a.split("=") for a in s.split())} if s.strip() else None
    for name, (w, h, n) in CONFIGS.items():
        net = api.EightLayersNet(w, h, options=opts)
        x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
        out = torch.empty((n,) + net.descs[-1].out_shape, dtype=torch.uint8, device="cuda")
        lat = torch.empty((n,) + net.descs[3].out_shape, dtype=torch.uint8, device="cuda")
        net.forward(x, out, lat)
        torch.cuda.synchronize()
        key = (out.cpu().numpy().tobytes(), lat.cpu().numpy().tobytes())
        same = ref.setdefault(name, key) == key
        g = net.capture(x, out, lat)
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 200
        net.profile(True)
        net.layer_ms(reset=True)
        for _ in range(30):
            net.forward(x, out, lat)
        ms, cnt = net.layer_ms()
        net.profile(False)
        per = " ".join(f"{1e3 * m / max(c, 1):5.1f}" for m, c in zip(ms, cnt))
        print(f"[{s or 'defaults':24s}] {name:8s} graph {t_graph * 1e6:7.1f} us  layers {per}  sum {1e3 * sum(m / max(c, 1) for m, c in zip(ms, cnt)):6.1f}  same={same}", flush=True)
