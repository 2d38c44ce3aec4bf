#!/usr/bin/env python3
"""Where does a wave of the pipelined conv kernel (layer 1, 8 x 4K) spend its cycles?  Needs the diagnostic build
(make -C simple_image_compression_network_amd/csrc EXTRA=-DSICN_STAMP=1 OUT=../../gpurun_build/libsicn_stamp.so
OBJDIR=../../gpurun_build/stamp_obj), which stamps s_memtime at the start of a workgroup, around its pass loop and at its end,
and records the CU it ran on.
usage: SICN_LIB=gpurun_build/libsicn_stamp.so [SICN_X_LDS_PAD=40000] [SICN_X_STAGGER=cycles] python tools/pass_stamps.py
  SICN_X_LDS_PAD  bytes of unused LDS per workgroup: 40000 leaves room for ONE workgroup per CU (a wave's solo pass rate)
  SICN_X_STAGGER  workgroups 256..511 (the second residents of the CUs) start that many cycles late"""
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
a0 = net.run_layers(0, 0, x)[0]
d1 = net.descs[1]
tiles = ((d1.OFM_ROW + 31) // 32) * ((d1.OFM_COL + 7) // 8) * n
slots = (tiles + 7) // 8 * 8 + 64
NF = 8
buf = torch.zeros((slots * 4 * NF,), dtype=torch.int64, device="cuda")
L.sicn_debug_stamp_buffer.argtypes = [ctypes.c_void_p]
L.sicn_debug_stagger.argtypes = [ctypes.c_ulonglong]
assert L.sicn_debug_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
assert L.sicn_debug_stagger(int(os.environ.get("SICN_X_STAGGER", "0"))) == 0
out1 = torch.empty((n,) + d1.out_shape, dtype=torch.uint8, device="cuda")
for _ in range(3):
    net.run_layers(1, 1, a0, out=out1)
torch.cuda.synchronize()
buf.zero_()
t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
t0.record()
net.run_layers(1, 1, a0, out=out1)
t1.record()
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(slots, 4, NF)
ids = np.nonzero(raw[:, 0, 0] > 0)[0]
s = raw[ids]
print(f"layer 1 alone (NHWC in / out): {t0.elapsed_time(t1):.3f} ms, {len(s)} workgroups stamped of {tiles} tiles")
tot, pro, loop, _, epi = (s[:, :, k].astype(np.float64) for k in range(5))
passes = 50
print(f"cycles per wave (mean over waves): whole life {tot.mean():.0f} = prologue {pro.mean():.0f} + {passes} passes {loop.mean():.0f} "
      f"+ epilogue {epi.mean():.0f}")
print(f"per pass {loop.mean() / passes:.0f} cycles (a pass is 32 MFMAs = 512 pipe cycles per wave, two waves share a SIMD's pipe)")
for name, v in (("whole", tot), ("prologue", pro), ("loop", loop), ("epilogue", epi)):
    q = np.percentile(v, [5, 50, 95])
    print(f"  {name:9s} p5 {q[0]:8.0f}  p50 {q[1]:8.0f}  p95 {q[2]:8.0f}")
# which workgroups share a CU, and how far apart do they start?
hw, xcc = s[:, 0, 6].astype(np.int64), s[:, 0, 7].astype(np.int64) & 15
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
start = s[:, 0, 5].astype(np.float64)
end = start + tot[:, 0]
print("distinct CUs seen:", len(np.unique(cu)), " kernel span", int(end.max() - start.min()), "cycles")
first = {}
for b, c_, t in zip(ids, cu, start):
    if b < 1024:
        first.setdefault(int(c_), []).append((int(b), int(t - start.min())))
for v in list(first.values())[:5]:
    print("  one CU's first workgroups (blockIdx, start):", sorted(v)[:4])
order = np.argsort(start)
last, gaps = {}, []
for i in order:
    c_ = int(cu[i])
    if c_ in last:
        gaps.append(start[i] - last[c_])
    last[c_] = start[i]
gaps = np.array(gaps)
print(f"start-to-start gap between consecutive workgroups on one CU: p10 {np.percentile(gaps, 10):.0f}  p50 {np.percentile(gaps, 50):.0f}  "
      f"p90 {np.percentile(gaps, 90):.0f} cycles (life {tot.mean():.0f}; lockstep = gaps of ~0 and ~life alternating, a perfect stagger = life / 2)")
# how fast does a workgroup's loop run while its CU mate is ALSO in its loop, and how fast while the mate is in its prologue /
# epilogue (or absent)?  50 passes = shared_cycles / T_shared + alone_cycles / T_alone, least squares over all workgroups
t_loop = start + pro[:, 0]
t_loop_end = t_loop + loop[:, 0]
by_cu = {}
for i in range(len(s)):
    by_cu.setdefault(int(cu[i]), []).append(i)
shared = np.zeros(len(s))
for members in by_cu.values():
    members.sort(key=lambda i: t_loop[i])
    for a_i, i in enumerate(members):
        for j in members[max(0, a_i - 3):a_i + 4]:
            if j != i:
                shared[i] += max(0.0, min(t_loop_end[i], t_loop_end[j]) - max(t_loop[i], t_loop[j]))
alone = loop[:, 0] - shared
A = np.stack([shared, alone], axis=1)
coef, *_ = np.linalg.lstsq(A, np.full(len(s), float(passes)), rcond=None)
print(f"of a loop's {loop[:, 0].mean():.0f} cycles, {shared.mean():.0f} are shared with the CU mate's loop and {alone.mean():.0f} are not; "
      f"fit: {1 / coef[0]:.0f} cycles per pass while shared, {1 / coef[1]:.0f} while the mate is in its prologue / epilogue")
np.savez_compressed(ROOT / "gpurun_out" / "pass_stamps.npz", raw=s, ids=ids) if (ROOT / "gpurun_out").exists() else None
