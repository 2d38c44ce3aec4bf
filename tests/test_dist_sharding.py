"""N>1 host logic on CPU: world_size 2 over gloo.  The compute function injected here is the oracle
(allowed in tests); in production it is the HIP path, one process per GPU, backend nccl (= RCCL)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from simple_image_compression_network_amd.dist import broadcast_params, checksum, run_sharded, shard_indices

ROOT = Path(__file__).resolve().parent.parent
W, H, N = 32, 16, 5


def _make_image(i):
    return np.random.default_rng(i).integers(0, 256, (H, W, 3), dtype=np.uint8)


def _oracle_compute(batch):
    from oracle import sicn_ref
    params = sicn_ref.load_param_fixture(ROOT / "tests" / "golden" / "param_weights.npz")
    outs = [sicn_ref.eight_layers_net_ref(x, params) for x in batch]
    return np.stack([o[7] for o in outs]), np.stack([o[3] for o in outs])


def test_shard_indices_partition():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard_indices(64, 3, 8) == [3, 11, 19, 27, 35, 43, 51, 59]      # BASELINE config 4: 8 per GPU
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def test_single_process_table():
    table = run_sharded(N, _make_image, _oracle_compute)
    recon, latent = _oracle_compute(np.stack([_make_image(i) for i in range(N)]))
    assert table == {i: [checksum(recon[i]), checksum(latent[i])] for i in range(N)}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # start-up collective: rank 0's weight tables reach every rank
        from simple_image_compression_network_amd import api
        params = api.load_param_weights()
        want = [[t.m_weights.copy() for t in pair] for pair in params]
        if rank != 0:
            for pair in params:
                for t in pair:
                    t.m_weights[...] = 0
        broadcast_params(params, src=0)
        assert all(np.array_equal(t.m_weights, w) for pair, ws in zip(params, want) for t, w in zip(pair, ws))
        q.put((rank, run_sharded(N, _make_image, _oracle_compute)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo_agree_with_single_process():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = run_sharded(N, _make_image, _oracle_compute)
    assert results[0] == results[1] == single
