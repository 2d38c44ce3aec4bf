#!/usr/bin/env python3
"""Layer 0 + its GDN in one kernel (k_l0g) on 8 x 4K, alone: ms per launch (events).  SICN_LIB selects an experiment build
(tools/build_variant.sh NAME "-DSICN_EXP_L0G_..." k_l0g.hip).  usage: l0g_speed.py [gdn_fuse=0|1]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_image_compression_network_amd import api  # noqa: E402
from simple_image_compression_network_amd.hyperprior import random_gdn_params  # noqa: E402

fuse = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, w, h = 8, 3840, 2160
rng = np.random.default_rng(0)
d = api.eight_layer_descs(w, h)[0]
params = api.load_param_weights()[0]
beta, gamma = random_gdn_params(rng, 128)
g = api.GDN(beta, gamma, inverse=False, shift=12)
x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
out = torch.empty((n,) + d.out_shape, dtype=torch.uint8, device="cuda")
net = api.EightLayersNet(descs=[d], params=[params], gdn=[g], options={"gdn_fuse": fuse} if fuse else None)   # weights resident: only the launch is timed
net.workspace(n)
run = lambda: net.run_layers(0, 0, x, out=out)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(10):
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"{os.environ.get('SICN_LIB', 'libsicn.so').split('/')[-2:]} gdn_fuse={fuse}: layer 0 + GDN on 8 x 4K: min {min(ts):.3f} ms, median {sorted(ts)[5]:.3f} ms")
