"""Image sharding over the GPUs of one node.

Images are independent (conv_nonsquare_top.cpp:295-357 keeps no cross-image state), so the only
multi-GPU parallelism of this path is along the batch: image i belongs to rank i mod world.  There
is NO data-path collective on that axis.  The collectives here are bookkeeping (per-image checksums / timing) and
run over whatever backend the process group was created with: "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.  Weights (1.44 MB packed) are read by rank 0 and replicated with one
broadcast at start-up (`broadcast_params`), so every rank provably runs the same tables.

Optional second axis ("single-image spatial split"): ONE large image over the ranks by horizontal bands with a
recomputed halo (`band_plan`, `forward_banded`) — the only place on this path where tensors cross GPUs: the bands of the
reconstruction (and of the latent) are all-gathered.
"""
from __future__ import annotations

import os

import zlib
from typing import Callable, Dict, List, Sequence

import numpy as np

# the device-resident forms are the product; `run_sharded` / `forward_banded` (numpy in, numpy out, one host hop per image) stay
# importable as the vehicles of the oracle-injected CPU tests (tests/test_dist_sharding.py) but are not part of the public surface
__all__ = ["shard_indices", "checksum", "checksum_t", "run_sharded_tensors", "broadcast_params", "band_plan",
           "forward_banded_tensors", "BandedNet", "BAND_HALO"]


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


# ---- one image over several GPUs: horizontal bands with a recomputed halo ------------------------------------------------
# Output row y of eight_layers_net depends on latent rows [y/16 - 2, y/16 + 2] (four deconv522 layers, each reaching one
# input row to either side: 1 + 1/2 + 1/4 + 1/8 < 2 latent rows) and latent row r on input rows [16 r - 30, 16 r + 30] (four
# conv2d layers: 2 + 4 + 8 + 16), so rows [a, b) of the reconstruction are a function of input rows [a - 62, b + 62) only.
# A band that starts and ends on multiples of 16 keeps every layer's stride-2 grid aligned with the whole image's, and with 64
# rows of halo the zero padding the band sees at its cut edges (instead of the neighbouring rows) cannot reach the rows it keeps.
BAND_HALO = 64


def band_plan(height: int, n_bands: int, halo: int = BAND_HALO, align: int = 16):
    """Cut `height` input rows into `n_bands` bands: [(in_begin, in_end, keep_begin, keep_end)], all in input rows; keep ranges
    are multiples of `align` (but for the image's last row), partition [0, height) and are non-empty for every band; a band
    reads [in_begin, in_end) = its keep range widened by `halo` and clipped to the image."""
    if height <= 0 or n_bands <= 0 or halo % align or halo < 62:
        raise ValueError("need height, n_bands > 0 and a halo that is a multiple of the alignment and >= 62 rows")
    units = (height + align - 1) // align                  # latent rows
    if n_bands > units:
        raise ValueError(f"{n_bands} bands for {units} rows of {align}")
    plan = []
    for b in range(n_bands):
        k0 = align * (units * b // n_bands)
        k1 = min(height, align * (units * (b + 1) // n_bands))
        plan.append((max(0, k0 - halo), min(height, k1 + halo), k0, k1))
    return plan


def forward_banded(compute: Callable[[np.ndarray], Sequence[np.ndarray]], image: np.ndarray, n_bands: int = 0, group=None):
    """`eight_layers_net` of ONE image [H][W][3] computed band by band: `compute` maps a band [h][W][3] to (reconstruction
    [16 ceil(h/16)][W'][3], latent [ceil(h/16)][..][192]) — the HIP path in production, the oracle in the CPU tests.
    With an initialised process group rank r computes band r (n_bands = world size) and the kept rows are all-gathered
    (RCCL over xGMI with backend nccl); without one, the bands are computed one after the other (n_bands as given).
    Returns (reconstruction, latent) of the whole image on every rank — byte-identical to one call on the whole image."""
    import torch
    import torch.distributed as dist

    # SICN_FORCE_COLLECTIVES=1: take the collective path with ONE rank too (a 1-GPU box can then run the RCCL all-gather code)
    distributed = dist.is_available() and dist.is_initialized() and (
        dist.get_world_size(group) > 1 or os.environ.get("SICN_FORCE_COLLECTIVES") == "1")
    if distributed:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        n_bands = world
    else:
        rank, world = 0, 1
        if n_bands <= 0:
            raise ValueError("n_bands must be given without a process group")
    H = image.shape[0]
    plan = band_plan(H, n_bands)
    out_rows = 16 * ((H + 15) // 16)                        # the net's output height (conv rounds up, deconv doubles)

    def one(b):
        i0, i1, k0, k1 = plan[b]
        recon, latent = compute(np.ascontiguousarray(image[i0:i1]))
        last = b == n_bands - 1
        r1 = (out_rows if last else k1) - i0                 # the last band also owns the rows the rounding adds
        return (np.ascontiguousarray(recon[k0 - i0:r1]), np.ascontiguousarray(latent[(k0 - i0) // 16:(r1 + 15) // 16]))

    if not distributed:
        parts = [one(b) for b in range(n_bands)]
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
    mine = one(rank)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    outs = []
    for part, full_rows in ((mine[0], out_rows), (mine[1], (H + 15) // 16)):
        # bands differ in height by at most one row of 16: pad to the tallest, gather, cut
        rows = [((out_rows if b == n_bands - 1 else plan[b][3]) - plan[b][2]) for b in range(n_bands)]
        if full_rows != out_rows:
            rows = [(r + 15) // 16 for r in rows]
        tallest = max(rows)
        buf = np.zeros((tallest,) + part.shape[1:], np.uint8)
        buf[:part.shape[0]] = part
        mine_t = torch.from_numpy(buf).to(dev)
        gathered = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(gathered, mine_t, group=group)
        outs.append(np.concatenate([g.cpu().numpy()[:rows[b]] for b, g in enumerate(gathered)]))
    return outs[0], outs[1]


# ---- the same two axes on DEVICE tensors: nothing crosses PCIe ---------------------------------------------
def checksum_t(t):
    """Order-sensitive checksum of a byte tensor computed where the tensor lives (two modular sums, Fletcher style); returns a
    0-d int64 tensor on t's device.  Same value on CPU and GPU for the same bytes."""
    import torch
    v = t.reshape(-1).to(torch.int64)
    idx = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 65521
    return (v.sum() % 4294967291) * 65536 + ((v * idx).sum() % 65521)


def run_sharded_tensors(n_images: int, make_batch, compute, group=None):
    """`run_sharded` without the host hop: `make_batch(indices)` returns this rank's images as ONE tensor [n][H][W][3] on the
    compute device, `compute` maps it to (recon, latent) tensors on that device, the per-image checksums are taken on the
    device (`checksum_t`) and the table {global image index: [recon, latent]} is all-gathered on the process group's device
    (RCCL with backend nccl).  Returns the table as a dict on every rank."""
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    mine = shard_indices(n_images, rank, world)
    per_rank = (n_images + world - 1) // world
    backend = dist.get_backend(group) if world > 1 else None
    if mine:
        batch = make_batch(mine)
        recon, latent = compute(batch)
        rows = [torch.stack([torch.tensor(i, dtype=torch.int64, device=recon.device), checksum_t(recon[k]), checksum_t(latent[k])])
                for k, i in enumerate(mine)]
        table = torch.stack(rows)
        cdev = table.device if backend in (None, "nccl") else torch.device("cpu")
        table = table.to(cdev)
    else:
        cdev = (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
        table = torch.empty((0, 3), dtype=torch.int64, device=cdev)
    pad = torch.full((per_rank - table.shape[0], 3), -1, dtype=torch.int64, device=table.device)
    table = torch.cat([table, pad])
    if world == 1:
        gathered = table
    else:
        gathered = torch.empty((world * per_rank, 3), dtype=torch.int64, device=table.device)
        dist.all_gather_into_tensor(gathered, table, group=group)
    out: Dict[int, List[int]] = {}
    for idx, a, b in gathered.cpu().tolist():
        if idx >= 0:
            out[int(idx)] = [int(a), int(b)]
    if sorted(out) != list(range(n_images)):
        raise RuntimeError("sharding lost or duplicated images")
    return out


def forward_banded_tensors(compute, image, n_bands: int = 0, group=None):
    """`forward_banded` on tensors: `image` [H][W][3] uint8 on the compute device (the same bytes on every rank), `compute` maps a
    band tensor [h][W][3] to (recon [16 ceil(h/16)][W'][3], latent [ceil(h/16)][..][192]) tensors on that device.  Rank r
    computes band r; ONE all_gather_into_tensor per output moves the kept rows (padded to the tallest band) between the
    devices — with backend nccl straight from and to GPU memory over xGMI.  Returns (reconstruction, latent) of the whole
    image on every rank, byte-identical to one call on the whole image."""
    import torch
    import torch.distributed as dist

    # SICN_FORCE_COLLECTIVES=1: take the collective path with ONE rank too (a 1-GPU box can then run the RCCL all-gather code)
    distributed = dist.is_available() and dist.is_initialized() and (
        dist.get_world_size(group) > 1 or os.environ.get("SICN_FORCE_COLLECTIVES") == "1")
    if distributed:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        n_bands = world
    else:
        rank, world = 0, 1
        if n_bands <= 0:
            raise ValueError("n_bands must be given without a process group")
    H = image.shape[0]
    plan = band_plan(H, n_bands)
    out_rows = 16 * ((H + 15) // 16)

    def one(b):
        i0, i1, k0, k1 = plan[b]
        recon, latent = compute(image[i0:i1])                # a row range of a row-major tensor is contiguous: no copy
        r1 = (out_rows if b == n_bands - 1 else k1) - i0     # the last band also owns the rows the rounding adds
        return recon[k0 - i0:r1], latent[(k0 - i0) // 16:(r1 + 15) // 16]

    if not distributed:
        parts = [one(b) for b in range(n_bands)]
        return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    mine = one(rank)
    on_cpu = dist.get_backend(group) != "nccl"
    outs = []
    for part, unit in ((mine[0], 1), (mine[1], 16)):
        rows = [((out_rows if b == n_bands - 1 else plan[b][3]) - plan[b][2]) for b in range(n_bands)]
        rows = [(r + unit - 1) // unit for r in rows]
        tallest = max(rows)
        send = torch.zeros((tallest,) + tuple(part.shape[1:]), dtype=torch.uint8, device=part.device)
        send[:part.shape[0]] = part
        if on_cpu:
            send = send.cpu()
        recv = torch.empty((world * tallest,) + tuple(part.shape[1:]), dtype=torch.uint8, device=send.device)
        dist.all_gather_into_tensor(recv, send, group=group)
        outs.append(torch.cat([recv[b * tallest:b * tallest + rows[b]] for b in range(n_bands)]).to(part.device))
    return outs[0], outs[1]


class BandedNet:
    """ONE image of width x height over the ranks of the process group by horizontal bands: rank r owns an EightLayersNet of its
    band's height (the HIP path) and `forward(image)` returns the whole reconstruction [1][16 ceil(H/16)][W'][3] on every rank's
    device, byte-identical to the one-GPU result; only the kept rows cross xGMI (one RCCL all-gather)."""

    def __init__(self, width: int, height: int, world: int, rank: int, params=None, device=None, group=None):
        from . import api
        self.group, self.world, self.rank, self.height = group, world, rank, height
        i0, i1, _, _ = band_plan(height, world)[rank]
        self.net = api.EightLayersNet(width, i1 - i0, params=params, device=device)

    def forward(self, image, want_latent: bool = False):
        def compute(band):
            rec, lat = self.net.forward(band[None], want_latent=True)
            return rec[0], lat[0]
        rec, lat = forward_banded_tensors(compute, image[0], group=self.group)
        return (rec[None], lat[None]) if want_latent else rec[None]
