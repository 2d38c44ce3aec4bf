#!/usr/bin/env python3
"""Where does a wave of the pipelined DECONV kernel (layer 6, 8 x 4K, in the chain) spend its cycles?  Diagnostic build, see
tools/pass_stamps.py.  Stamps: workgroup start, loop start, then (passes done, epilogue done) for each of the 4 output phases."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("SICN_LIB", str(ROOT / "gpurun_build" / "libsicn_stamp.so"))
from simple_image_compression_network_amd import _lib, api  # noqa: E402

L = _lib.lib()
n, W, H = 8, 3840, 2160
net = api.EightLayersNet(W, H)
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, H, W, 3), dtype=np.uint8)).cuda()
a4 = net.run_layers(0, 4, x)[0]                       # layer 4's output (NHWC) = input of the stamped run: layers 5, 6
out6 = torch.empty((n,) + net.descs[6].out_shape, dtype=torch.uint8, device="cuda")
slots = 16320 + 64
buf = torch.zeros((slots * 4 * 12,), dtype=torch.int64, device="cuda")
L.sicn_debug_stamp_buffer.argtypes = [ctypes.c_void_p]
assert L.sicn_debug_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
for _ in range(10):
    net.run_layers(5, 6, a4, out=out6)                # layer 5 -> (PHASE layout) -> layer 6; layer 6 runs last: its stamps stay
torch.cuda.synchronize()
buf.zero_()
net.profile(True)
net.layer_ms(reset=True)
net.run_layers(5, 6, a4, out=out6)
torch.cuda.synchronize()
ms, cnt = net.layer_ms()
s = buf.cpu().numpy().reshape(slots, 4, 12)
s = s[s[:, 0, 0] > 0].astype(np.float64)
t = s[:, :, :10]
life = t[:, :, 9] - t[:, :, 0]
pro = t[:, :, 1] - t[:, :, 0]
print(f"layer 6 in the chain: {ms[6] / max(cnt[6], 1):.3f} ms; {len(s)} workgroups; life {life.mean():.0f} cycles = prologue {pro.mean():.0f}", end="")
passes = [18, 12, 12, 8]
tot_loop = tot_epi = 0.0
prev = t[:, :, 1]
for ph in range(4):
    loop = t[:, :, 2 + 2 * ph] - prev
    epi = t[:, :, 3 + 2 * ph] - t[:, :, 2 + 2 * ph]
    prev = t[:, :, 3 + 2 * ph]
    tot_loop += loop.mean()
    tot_epi += epi.mean()
    print(f" + phase {ph}: {passes[ph]} passes {loop.mean():.0f} ({loop.mean() / passes[ph]:.0f} per pass), epilogue {epi.mean():.0f}", end="")
print(f"\npasses {tot_loop:.0f} ({100 * tot_loop / life.mean():.0f} %), epilogues {tot_epi:.0f} ({100 * tot_epi / life.mean():.0f} %), "
      f"prologue {100 * pro.mean() / life.mean():.0f} %; the pipe's own time for a tile: 50 passes x 2 waves x 512 = 51200 per workgroup pair")
