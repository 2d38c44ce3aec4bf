"""host_pipeline.HostPipeline: host buffers in, host buffers out, three streams — the results must be those of the resident path
whatever the overlap does (distinct data in every batch, more batches than slots, slots re-used)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import sicn_ref  # noqa: E402

gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_host_pipeline_equals_resident_path_and_oracle(depth):
    import torch
    from simple_image_compression_network_amd import api
    from simple_image_compression_network_amd.host_pipeline import HostPipeline
    w, h, n, batches = 208, 112, 3, 7
    net = api.EightLayersNet(w, h)
    hp = HostPipeline(net, n, depth=depth)
    rng = np.random.default_rng(depth)
    h_in = [HostPipeline.pinned((n, h, w, 3)) for _ in range(batches)]
    for t in h_in:
        t.copy_(torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)))
    h_out = [HostPipeline.pinned((n,) + net.descs[-1].out_shape) for _ in range(batches)]
    h_lat = [HostPipeline.pinned((n,) + net.descs[3].out_shape) for _ in range(batches)]
    for t in h_out + h_lat:
        t.fill_(0xEE)
    hp.run(h_in, h_out, h_lat)
    hp.synchronize()
    for i in range(batches):
        out, lat = net.forward(h_in[i].cuda())
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), h_out[i]), f"batch {i}: reconstruction differs from the resident path"
        assert torch.equal(lat.cpu(), h_lat[i]), f"batch {i}: latent differs from the resident path"
    # and the resident path is the oracle's (one image is enough here: the chain itself is covered elsewhere)
    z = np.load(ROOT / "tests/golden/param_weights.npz")
    from simple_image_compression_network_amd.config import eight_layer_descs
    descs = eight_layer_descs(w, h)
    params = [(sicn_ref.unpack_finn_tiles(z[f"w{k}_words"], d.SIMD, d.PE, d.IFM_CH, d.OFM_CH), z[f"b{k}"], d.transposed)
              for k, d in enumerate(descs)]
    outs = sicn_ref.eight_layers_net_ref(h_in[batches - 1][0].numpy(), params)
    assert np.array_equal(h_out[batches - 1][0].numpy(), outs[7])
    assert np.array_equal(h_lat[batches - 1][0].numpy(), outs[3])
    # a second run re-uses the slots (events of the first run still pending semantics)
    hp.run(h_in[:2], h_out[:2], h_lat[:2])
    hp.synchronize()
    out, _ = net.forward(h_in[1].cuda())
    assert torch.equal(out.cpu(), h_out[1])


@gpu
def test_host_pipeline_rejects_pageable_and_mismatched_buffers():
    import torch
    from simple_image_compression_network_amd import api
    from simple_image_compression_network_amd.host_pipeline import HostPipeline
    net = api.EightLayersNet(64, 32)
    hp = HostPipeline(net, 1)
    good_in, good_out = HostPipeline.pinned((1, 32, 64, 3)), HostPipeline.pinned((1,) + net.descs[-1].out_shape)
    with pytest.raises(TypeError):
        hp.run([torch.zeros((1, 32, 64, 3), dtype=torch.uint8)], [good_out])          # pageable input
    with pytest.raises(ValueError):
        hp.run([good_in, good_in], [good_out])                                        # one output per input
    with pytest.raises(ValueError):
        HostPipeline(net, 1, depth=0)
