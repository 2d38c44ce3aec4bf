#!/usr/bin/env python3
"""SHA-256 of what the hyperprior configuration (BASELINE.json configs[4]) produces for bench.py's synthetic 4K image 0, computed by the
ORACLE's stage-by-stage statement (oracle/hyper_pipeline.py: C closed-form layers, oracle/sicn_gdn_oracle.c, the two container oracles)
with the PARAM main weights and HyperpriorCodec's seed-0 hyper / GDN parameters (hyperprior.hyper_parameters).  bench.py's hyperprior leg
compares the GPU's latent, both containers and the reconstruction of image 0 with these (VERDICT r4 item 2 iii: it used to compare the
kernels only with themselves).  PARITY UNPINNED (no reference counterpart); GDN specification version 2.

Image g: numpy default_rng(g).integers(0, 256, (2160, 3840, 3), uint8).  Writes tests/golden/hyper_4k_hashes.json
{"<g>": {"y": sha, "z_container": sha, "y_container": sha, "recon": sha, "z_bytes": n, "y_bytes": n}, "gdn_spec_version": 2}.
~1.5 min per image on 8 cores.   usage: make_hyper_hashes.py [first [last]]   (default 0 1)"""
import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from oracle.hyper_pipeline import hyper_pipeline_ref  # noqa: E402
from simple_image_compression_network_amd.codec import auto_stream_symbols  # noqa: E402
from simple_image_compression_network_amd.config import eight_layer_descs  # noqa: E402
from simple_image_compression_network_amd.hyperprior import hyper_parameters  # noqa: E402

W, H = 3840, 2160
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out_path = HERE / "hyper_4k_hashes.json"
z = np.load(HERE / "param_weights.npz")
words, bias = [z[f"w{n}_words"] for n in range(8)], [z[f"b{n}"] for n in range(8)]
descs = eight_layer_descs(W, H)
hp = hyper_parameters(W, H, seed=0)
zh, zw, zc = hp["da"][-1].out_shape
table = {"gdn_spec_version": 2}
sha = lambda b: hashlib.sha256(b if isinstance(b, bytes) else np.ascontiguousarray(b).tobytes()).hexdigest()
for g in range(first, last):
    x = np.random.default_rng(g).integers(0, 256, (H, W, 3), dtype=np.uint8)
    t0 = time.time()
    r = hyper_pipeline_ref(x, descs, words, bias, hp, (W, H), auto_stream_symbols(zh * zw * zc))
    table[str(g)] = {"y": sha(r["y"]), "z_container": sha(r["z_container"]), "y_container": sha(r["y_container"]), "recon": sha(r["recon"]),
                     "z_bytes": len(r["z_container"]), "y_bytes": len(r["y_container"])}
    print(f"image {g}: {time.time() - t0:.1f}s", flush=True)
out_path.write_text(json.dumps(table, indent=1, sort_keys=True) + "\n")
