#!/usr/bin/env python3
"""Run by tests/test_gpu_parity.py::test_alt_build_32x32x32_kernels in a process of its own with SICN_LIB = libsicn_alt.so:
the 32x32x32 MFMA kernels of k_mfma.hip (sicn_options.mfma_shape = 32) against the oracle, standalone on every shape they are
instantiated for and inside a chain (internal layouts) against the Appendix-A hashes."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import sicn_ref  # noqa: E402
from simple_image_compression_network_amd import _lib, api  # noqa: E402
from simple_image_compression_network_amd.config import LayerDesc  # noqa: E402

assert _lib.lib().sicn_has_alt_kernels() == 1, "not the ALT build"
CASES = [(128, 128, 8, 16, 66, 18, 0), (128, 128, 8, 16, 7, 5, 0), (128, 128, 8, 16, 1, 1, 0), (128, 128, 4, 32, 131, 33, 0),
         (128, 192, 8, 24, 40, 22, 0), (128, 192, 8, 24, 65, 3, 0),
         (192, 128, 12, 16, 33, 9, 1), (192, 128, 12, 16, 3, 2, 1), (192, 128, 12, 16, 70, 17, 1),
         (128, 128, 8, 16, 34, 10, 1), (128, 128, 8, 16, 1, 1, 1), (128, 128, 8, 16, 65, 19, 1)]
for n, (cin, cout, simd, pe, w, h, tr) in enumerate(CASES):
    ow, oh = (2 * w, 2 * h) if tr else ((w + 1) // 2, (h + 1) // 2)
    d = LayerDesc(IFM_CH=cin, IFM_ROW=w, IFM_COL=h, OFM_CH=cout, OFM_ROW=ow, OFM_COL=oh, SIMD=simd, PE=pe,
                  W_TILES=(cout // pe) * (25 * cin // simd), transposed=tr)
    rng = np.random.default_rng(100 + n)
    W = rng.integers(-8, 8, (cout, 5, 5, cin)).astype(np.int8)
    b = rng.integers(-128, 128, cout).astype(np.int8)
    x = rng.integers(0, 128, (2,) + d.in_shape, dtype=np.uint8)
    fpw = api.FixedPointWeights(simd, 4, pe, d.W_TILES, sicn_ref.pack_finn_tiles(W, simd, pe))
    fn = api.deconv522 if tr else api.conv2d
    got = fn(d, fpw, b, torch.from_numpy(x).cuda(), None, 2, options={"mfma_shape": 32}).cpu().numpy()
    ref_fn = sicn_ref.deconv522_ref if tr else sicn_ref.conv2d_ref
    for i in range(2):
        assert np.array_equal(got[i], ref_fn(x[i], W, b)), (cin, cout, w, h, tr, i)
hashes = json.loads((ROOT / "tests" / "golden" / "appendix_a_hashes.json").read_text())
x = np.random.default_rng(0).integers(0, 256, (1, 256, 256, 3), dtype=np.uint8)
net = api.EightLayersNet(256, 256, options={"mfma_shape": 32})
out, latent = net.forward(torch.from_numpy(x).cuda())
torch.cuda.synchronize()
assert hashlib.sha256(out[0].cpu().numpy().tobytes()).hexdigest() == hashes["layers"]["rng256"][7]
assert hashlib.sha256(latent[0].cpu().numpy().tobytes()).hexdigest() == hashes["layers"]["rng256"][3]
print("alt kernels ok:", len(CASES), "layers + a 256x256 chain bit-exact on the 32x32x32 kernels")
