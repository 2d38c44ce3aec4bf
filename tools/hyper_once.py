"""Six hyperprior steps (encode + decode of 8 x 4K) for rocprofv3 --kernel-trace --stats: python3 tools/hyper_once.py "gdn_fuse=1" (sicn_options as k=v,k=v)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_image_compression_network_amd.hyperprior import HyperpriorCodec
spec = sys.argv[1] if len(sys.argv) > 1 else ""
opt = {k: int(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv)} or None
n, w, h = 8, 3840, 2160
x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
out = torch.empty_like(x)
hc = HyperpriorCodec(w, h, n, seed=0, options=opt)
for _ in range(6):
    hc.encode(x); hc.decode(out)
hc.check(); torch.cuda.synchronize()
