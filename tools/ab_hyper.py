"""A/B of the hyperprior step (8 x 4K) with sicn_options given on the command line: python ab_hyper.py "" "gdn_fuse=1" """
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_image_compression_network_amd.hyperprior import HyperpriorCodec

def parse(s):
    return {k: int(v) for k, v in (kv.split("=") for kv in s.split(",") if kv)} or None

specs = sys.argv[1:] or ["", "gdn_fuse=1"]
n, w, h = 8, 3840, 2160
x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
out = torch.empty_like(x)
res = {}
for rnd in range(3):
    for spec in specs:
        hc = HyperpriorCodec(w, h, n, seed=0, options=parse(spec))
        def step():
            hc.encode(x); hc.decode(out)
        step(); hc.check(); torch.cuda.synchronize()
        for _ in range(2): step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        res.setdefault(spec, []).append(min(ts))
        h0 = int(out.view(-1)[::9973].sum().item())
        print(f"round {rnd} [{spec or 'defaults'}] {min(ts):.3f} ms  bytes {sum(hc.bytes_per_image())} check {h0}", flush=True)
        del hc
for spec, v in res.items():
    print(f"[{spec or 'defaults'}] min {min(v):.3f} ms  median {sorted(v)[len(v)//2]:.3f}")
