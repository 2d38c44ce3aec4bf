#!/usr/bin/env python3
"""Where does a workgroup of the pipelined DECONV kernel spend its cycles on a SMALL grid (one image: the chip is at most
just filled)?  Diagnostic build (-DSICN_STAMP=1: k_deconv_p stamps s_memtime at workgroup start, loop start and behind the
passes / the epilogue of each of the 4 output phases).  usage: deconv_stamps_small.py WIDTH HEIGHT LAYER [sicn_options k=v ...]
  SICN_LIB=$PWD/gpurun_build/stamp/libsicn.so python tools/deconv_stamps_small.py 1920 1080 5"""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from simple_image_compression_network_amd import _lib, api  # noqa: E402

W, H, LAYER = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
opts = {k: int(v) for k, v in (a.split("=") for a in sys.argv[4:])}
L = _lib.lib()
net = api.EightLayersNet(W, H, options=opts or None)
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1, H, W, 3), dtype=np.uint8)).cuda()
a_in = net.run_layers(0, LAYER - 1, x)[0]
out = torch.empty((1,) + net.descs[LAYER].out_shape, dtype=torch.uint8, device="cuda")
slots = 8192
buf = torch.zeros((slots * 4 * 12,), dtype=torch.int64, device="cuda")
L.sicn_debug_stamp_buffer.argtypes = [ctypes.c_void_p]
assert L.sicn_debug_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
for _ in range(10):
    net.run_layers(LAYER, LAYER, a_in, out=out)
torch.cuda.synchronize()
buf.zero_()
net.profile(True)
net.layer_ms(reset=True)
net.run_layers(LAYER, LAYER, a_in, out=out)
torch.cuda.synchronize()
ms, cnt = net.layer_ms()
s = buf.cpu().numpy().reshape(slots, 4, 12)
s = s[s[:, 0, 0] > 0].astype(np.float64)
t = s[:, :, :10]
t0 = t[:, :, 0].min()
life = t[:, :, 9] - t[:, :, 0]
pro = t[:, :, 1] - t[:, :, 0]
d = net.descs[LAYER]
nq = d.IFM_CH // 32
taps = [9, 6, 6, 4]
print(f"layer {LAYER} of {W}x{H} alone: {1e3 * ms[LAYER] / max(cnt[LAYER], 1):.1f} us; {len(s)} workgroups; first start .. last end "
      f"{(t[:, :, 9].max() - t0):.0f} cycles; starts spread over {(t[:, :, 0].max() - t0):.0f}; life {life.mean():.0f} (max {life.max():.0f}) = prologue {pro.mean():.0f}")
prev = t[:, :, 1]
for ph in range(4):
    loop = t[:, :, 2 + 2 * ph] - prev
    epi = t[:, :, 3 + 2 * ph] - t[:, :, 2 + 2 * ph]
    prev = t[:, :, 3 + 2 * ph]
    npass = taps[ph] * nq // 2
    print(f"  phase {ph}: {npass} passes {loop.mean():.0f} cycles ({loop.mean() / npass:.0f} per pass), epilogue {epi.mean():.0f}")
