import sys, time, torch, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
from simple_image_compression_network_amd.hyperprior import HyperpriorCodec
n, w, h = 8, 3840, 2160
x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
out = torch.empty_like(x)
hc = HyperpriorCodec(w, h, n, seed=0)
for _ in range(3):
    hc.encode(x); hc.decode(out)
hc.check(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    hc.encode(x); hc.decode(out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"{os.environ.get('SICN_LIB','libsicn.so')}: hyperprior {dt*1e3:.3f} ms per step, {8.0*sum(hc.bytes_per_image())/(n*w*h):.4f} bit/pixel, round trip {bool(torch.equal(hc.y_hat, hc.y))}")
