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


def test_bench_uses_the_documented_partition():
    """bench.py deals the job's images out exactly as dist.shard_indices / DESIGN.md section 6 say (image i -> rank i mod world), so
    the one multi-GPU run the driver makes exercises the documented partition and looks its golden hashes up under the GLOBAL index
    (up to round 4 rank r took the contiguous block r B .. r B + B - 1)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.bench_image_ids(0, 1, 8) == list(range(8))                      # N = 1: images 0..7, as before
    assert bench.bench_image_ids(3, 8, 8) == [3, 11, 19, 27, 35, 43, 51, 59]     # BASELINE configs[3]: 64 images over 8 GPUs
    for world in (2, 4, 8):
        ids = [bench.bench_image_ids(r, world, 8) for r in range(world)]
        assert sorted(sum(ids, [])) == list(range(8 * world)) and all(i % world == r for r, part in enumerate(ids) for i in part)
    import json
    golden = json.loads((ROOT / "tests" / "golden" / "bench_4k_hashes.json").read_text())
    assert all(str(i) in golden for i in range(64))                              # every index any rank of an 8-GPU run looks up


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
        # the global image indices each rank owns, all-gathered: together 0 .. N-1, rank r holding exactly the i with i mod world == r
        mine = shard_indices(N, rank, world)
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        assert everyone[rank] == mine and all(i % world == r for r, part in enumerate(everyone) for i in part)
        assert sorted(sum(everyone, [])) == list(range(N))
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


# ---- one image over several ranks: bands with a recomputed halo --------------------------------------------------------------
from simple_image_compression_network_amd.dist import BAND_HALO, band_plan, forward_banded  # noqa: E402

BW, BH = 32, 304        # 19 latent rows: bands of uneven height, a last band that is not a multiple of 16 rows high


def _oracle_band(band):
    from oracle import c_oracle
    from simple_image_compression_network_amd.config import eight_layer_descs
    z = np.load(ROOT / "tests" / "golden" / "param_weights.npz")
    outs = c_oracle.run_net(eight_layer_descs(band.shape[1], band.shape[0]), [z[f"w{n}_words"] for n in range(8)],
                            [z[f"b{n}"] for n in range(8)], band, "direct", threads=2)
    return outs[7], outs[3]


def test_band_plan_partitions_rows():
    for height in (16, 17, 304, 1080, 2160):
        units = (height + 15) // 16
        for n in (1, 2, 3, 8):
            if n > units:
                with pytest.raises(ValueError):
                    band_plan(height, n)
                continue
            plan = band_plan(height, n)
            assert plan[0][2] == 0 and plan[-1][3] == height
            for (i0, i1, k0, k1), nxt in zip(plan, plan[1:] + [None]):
                assert 0 <= i0 <= k0 < k1 <= i1 <= height and k0 % 16 == 0 and i0 % 16 == 0
                assert k0 - i0 in (0, BAND_HALO) or i0 == 0
                if nxt is not None:
                    assert k1 == nxt[2] and k1 % 16 == 0
    with pytest.raises(ValueError):
        band_plan(304, 2, halo=48)       # fewer than the 62 rows the receptive field needs


def test_banded_forward_equals_whole_image():
    img = np.random.default_rng(11).integers(0, 256, (BH, BW, 3), dtype=np.uint8)
    whole_recon, whole_latent = _oracle_band(img)
    for n in (2, 3):
        recon, latent = forward_banded(_oracle_band, img, n_bands=n)
        assert np.array_equal(recon, whole_recon) and np.array_equal(latent, whole_latent), n
    odd = img[:297]                      # 297 rows: the net rounds up to 304 output rows; the last band owns them
    whole_recon, whole_latent = _oracle_band(odd)
    recon, latent = forward_banded(_oracle_band, odd, n_bands=2)
    assert recon.shape == whole_recon.shape and np.array_equal(recon, whole_recon) and np.array_equal(latent, whole_latent)


def _band_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        img = np.random.default_rng(11).integers(0, 256, (BH, BW, 3), dtype=np.uint8)
        recon, latent = forward_banded(_oracle_band, img)
        q.put((rank, checksum(recon), checksum(latent), recon.shape, latent.shape))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo_banded_image():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_band_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    img = np.random.default_rng(11).integers(0, 256, (BH, BW, 3), dtype=np.uint8)
    recon, latent = _oracle_band(img)
    for _, cr, cl, sr, sl in results:
        assert (cr, cl, sr, sl) == (checksum(recon), checksum(latent), recon.shape, latent.shape)


# ---- the tensor forms (device-resident in production; CPU tensors + gloo here) ------------------------------------------------
from simple_image_compression_network_amd.dist import checksum_t, forward_banded_tensors, run_sharded_tensors  # noqa: E402


def _oracle_band_t(band_t):
    import torch
    recon, latent = _oracle_band(band_t.numpy())
    return torch.from_numpy(np.ascontiguousarray(recon)), torch.from_numpy(np.ascontiguousarray(latent))


def _oracle_compute_t(batch_t):
    import torch
    recon, latent = _oracle_compute(batch_t.numpy())
    return torch.from_numpy(np.ascontiguousarray(recon)), torch.from_numpy(np.ascontiguousarray(latent))


def _make_batch_t(indices):
    import torch
    return torch.from_numpy(np.stack([_make_image(i) for i in indices]))


def test_tensor_forms_single_process():
    import torch
    img = np.random.default_rng(11).integers(0, 256, (BH, BW, 3), dtype=np.uint8)
    whole_recon, whole_latent = _oracle_band(img)
    recon, latent = forward_banded_tensors(_oracle_band_t, torch.from_numpy(img), n_bands=3)
    assert np.array_equal(recon.numpy(), whole_recon) and np.array_equal(latent.numpy(), whole_latent)
    table = run_sharded_tensors(N, _make_batch_t, _oracle_compute_t)
    recon_all, latent_all = _oracle_compute(np.stack([_make_image(i) for i in range(N)]))
    for i in range(N):
        assert table[i] == [int(checksum_t(torch.from_numpy(recon_all[i]))), int(checksum_t(torch.from_numpy(latent_all[i])))]
    a = torch.arange(1000, dtype=torch.uint8)
    assert int(checksum_t(a)) != int(checksum_t(a.flip(0)))       # order-sensitive


def _tensor_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        img = torch.from_numpy(np.random.default_rng(11).integers(0, 256, (BH, BW, 3), dtype=np.uint8))
        recon, latent = forward_banded_tensors(_oracle_band_t, img)
        q.put((rank, checksum(recon.numpy()), checksum(latent.numpy()), run_sharded_tensors(N, _make_batch_t, _oracle_compute_t)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo_tensor_forms():
    """world size 2: the band split and the image sharding on tensors (one all_gather_into_tensor each), against one process."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tensor_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    img = np.random.default_rng(11).integers(0, 256, (BH, BW, 3), dtype=np.uint8)
    recon, latent = _oracle_band(img)
    single = run_sharded_tensors(N, _make_batch_t, _oracle_compute_t)
    for _, cr, cl, table in results:
        assert (cr, cl) == (checksum(recon), checksum(latent)) and table == single
