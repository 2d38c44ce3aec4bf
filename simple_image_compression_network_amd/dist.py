"""Image sharding over the GPUs of one node.

Images are independent (conv_nonsquare_top.cpp:295-357 keeps no cross-image state), so the only
multi-GPU parallelism of this path is along the batch: image i belongs to rank i mod world.  There
is NO data-path collective.  The collectives here are bookkeeping (per-image checksums / timing) and
run over whatever backend the process group was created with: "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.  Weights (1.44 MB packed) are read by rank 0 and replicated with one
broadcast at start-up (`broadcast_params`), so every rank provably runs the same tables.
"""
from __future__ import annotations

import zlib
from typing import Callable, Dict, List, Sequence

import numpy as np

__all__ = ["shard_indices", "checksum", "run_sharded", "broadcast_params"]


def shard_indices(n_images: int, rank: int, world: int) -> List[int]:
    """Global image indices owned by `rank`: i with i % world == rank."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_images, world))


def checksum(arr: np.ndarray) -> int:
    """Order-sensitive 32-bit checksum of a byte tensor (adler32 of the raw bytes)."""
    return zlib.adler32(np.ascontiguousarray(arr).view(np.uint8).reshape(-1)) & 0xFFFFFFFF


def broadcast_params(params, src: int = 0, group=None, device=None):
    """One bookkeeping collective at start-up: the `m_weights` words of every
    (weights, bias) table of `params` (api.load_param_weights() format) are replaced IN PLACE by rank
    `src`'s.  Ranks other than `src` only need tables of the right geometry (e.g. zeros).  No-op without
    an initialised process group."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return params
    backend = dist.get_backend(group)
    dev = device if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    tables = [t for pair in params for t in pair]
    flat = np.concatenate([t.m_weights.reshape(-1) for t in tables]).view(np.uint8)     # uint64 words as bytes
    buf = torch.from_numpy(flat.copy()).to(dev)
    dist.broadcast(buf, src=src, group=group)
    words = buf.cpu().numpy().view(np.uint64)
    pos = 0
    for t in tables:
        n = t.m_weights.size
        t.m_weights[...] = words[pos:pos + n].reshape(t.m_weights.shape)
        pos += n
    return params


def run_sharded(n_images: int, make_image: Callable[[int], np.ndarray],
                compute: Callable[[np.ndarray], Sequence[np.ndarray]], group=None) -> Dict[int, List[int]]:
    """Every rank runs `compute` (a batch [n][H][W][3] -> (recon, latent) as numpy arrays) on its own
    images and the per-image checksums of (recon, latent) are all-gathered, so that every rank ends
    with the full table {global image index: [recon checksum, latent checksum]}.

    `compute` is the HIP path in production (api.EightLayersNet.forward wrapped to numpy); the CPU
    tests inject the oracle here — this module itself never imports it."""
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    mine = shard_indices(n_images, rank, world)
    per_rank = (n_images + world - 1) // world
    table = np.full((per_rank, 3), -1, dtype=np.int64)          # (index, recon, latent), -1 = padding
    if mine:
        batch = np.stack([make_image(i) for i in mine])
        recon, latent = compute(batch)
        for k, i in enumerate(mine):
            table[k] = (i, checksum(recon[k]), checksum(latent[k]))
    if world == 1:
        gathered = [torch.from_numpy(table)]
    else:
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        t = torch.from_numpy(table).to(dev)
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t, group=group)
    out: Dict[int, List[int]] = {}
    for g in gathered:
        for idx, a, b in g.cpu().numpy().tolist():
            if idx >= 0:
                out[int(idx)] = [int(a), int(b)]
    if sorted(out) != list(range(n_images)):
        raise RuntimeError("sharding lost or duplicated images")
    return out
