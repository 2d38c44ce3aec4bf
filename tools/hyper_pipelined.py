#!/usr/bin/env python3
"""Hyperprior steps (8 x 4K) one after the other on one stream against two batches in flight: the ENCODE of batch k + 1 on one stream
beside the DECODE of batch k on another (two codec objects, events for the hand-over), the way a service that both encodes and decodes
would run them.  The coders' kernels are latency-bound chains on a few waves per SIMD; beside a transform kernel of the other job they
cost little.  usage: hyper_pipelined.py [steps=12]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_image_compression_network_amd.hyperprior import HyperpriorCodec  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n, w, h = 8, 3840, 2160
x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
codecs = [HyperpriorCodec(w, h, n, seed=0) for _ in range(2)]
outs = [torch.empty_like(x) for _ in range(2)]


def serial(k):
    hc = codecs[0]
    for _ in range(k):
        hc.encode(x)
        hc.decode(outs[0])


def pipelined(k, se, sd):
    enc_done = [torch.cuda.Event() for _ in range(2)]
    dec_done = [torch.cuda.Event() for _ in range(2)]
    for i in range(k):
        hc = codecs[i & 1]
        with torch.cuda.stream(se):
            if i >= 2:
                se.wait_event(dec_done[i & 1])      # this object's buffers are free again
            hc.encode(x)
            enc_done[i & 1].record(se)
        with torch.cuda.stream(sd):
            sd.wait_event(enc_done[i & 1])
            hc.decode(outs[i & 1])
            dec_done[i & 1].record(sd)


def timed(fn):
    fn(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


t_serial = timed(serial)
ref = outs[0].clone()
# optional: stream priorities of the encode / decode side (-1 = high), e.g. "hyper_pipelined.py 12 0 -1"
pe, pd = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 0)
se, sd = torch.cuda.Stream(priority=pe), torch.cuda.Stream(priority=pd)
t_pipe = timed(lambda k: pipelined(k, se, sd))
for hc in codecs:
    hc.check()
ok = bool(torch.equal(outs[0], ref)) and bool(torch.equal(outs[1], ref))
print(f"hyperprior 8 x 4K: one stream {t_serial:.3f} ms per step; encode(k+1) beside decode(k) on two streams (priorities {pe} / {pd}) {t_pipe:.3f} ms per step; outputs equal {ok}")
t_serial2 = timed(serial)
print(f"one stream again {t_serial2:.3f} ms")
