#!/usr/bin/env python3
"""SHA-256 of the latent (layer 3) and the reconstruction (layer 7) of bench.py's synthetic 4K images, computed by
the ORACLE's direct closed form (oracle/sicn_oracle.c: sicn_or_layer_direct, itself pinned to reference-made vectors
by tests/test_oracle_golden.py) with the PARAM weights.  bench.py compares the GPU's timed outputs with these after the
timed loop and reports `output_bit_exact`.

Image g (global index = rank * 8 + i, as bench.py seeds them): numpy default_rng(g).integers(0, 256, (2160, 3840, 3), uint8).
Writes tests/golden/bench_4k_hashes.json {"<g>": [sha256(latent), sha256(out)]}.  ~15 s per image on 8 cores.

usage: make_bench_hashes.py [first [last]]   (default 0 64); existing entries are kept.
"""
import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from oracle import c_oracle  # noqa: E402
from simple_image_compression_network_amd.config import eight_layer_descs  # noqa: E402

W, H = 3840, 2160
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 64
out_path = HERE / "bench_4k_hashes.json"
table = json.loads(out_path.read_text()) if out_path.exists() else {}
z = np.load(HERE / "param_weights.npz")
words, bias = [z[f"w{n}_words"] for n in range(8)], [z[f"b{n}"] for n in range(8)]
descs = eight_layer_descs(W, H)
import os
threads = int(os.environ.get("ORACLE_THREADS", os.cpu_count() or 8))
for g in range(first, last):
    if str(g) in table:
        continue
    x = np.random.default_rng(g).integers(0, 256, (H, W, 3), dtype=np.uint8)
    t0 = time.time()
    outs = c_oracle.run_net(descs, words, bias, x, "direct", threads=threads)
    table[str(g)] = [hashlib.sha256(outs[3].tobytes()).hexdigest(), hashlib.sha256(outs[7].tobytes()).hexdigest()]
    out_path.write_text(json.dumps({k: table[k] for k in sorted(table, key=int)}, indent=0) + "\n")
    print(f"image {g}: {time.time() - t0:.1f}s", flush=True)
