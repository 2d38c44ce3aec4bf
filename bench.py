#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the 8-layer integer transform (encode L0-L3 + decode L4-L7,
conv_nonsquare_top.cpp:295-357) on synthetic 4K RGB images, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over this rank's batch of IMAGES_PER_GPU 3840x2160x3 uint8
images, already resident in HBM (BASELINE.json configs[3]: 64 images over 8 GPUs = 8 per GPU; weak
scaling: every rank always owns 8 images, no data-path collective — images are independent).
Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel, timed live with hipEvents
recorded by the library on the launch stream (sicn_net_profile); `cpu_baseline` is the oracle's
dataflow-faithful C port (the stand-in for the reference's HLS C-simulation, which needs Vivado
headers) timed on this host on a bounded crop of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_INT8_TOPS = 5000.0   # dense int8 MFMA = 2x the ~2.5 PFLOP/s bf16 dense peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0     # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--images-per-gpu", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, nargs=2, default=[256, 256], metavar=("W", "H"))
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) is the real thing; gloo lets several ranks share one GPU "
                         "to rehearse the multi-rank path on a 1-GPU box")
    return ap.parse_args()


def cpu_baseline(image0: np.ndarray, sample_wh):
    """Oracle (test infrastructure) used ONLY as the reported CPU baseline: stage-by-stage dataflow
    port, one thread, on a crop of the first synthetic image."""
    from oracle import c_oracle
    from simple_image_compression_network_amd.config import eight_layer_descs
    w, h = sample_wh
    z = np.load(ROOT / "tests" / "golden" / "param_weights.npz")
    crop = np.ascontiguousarray(image0[:h, :w])
    descs = eight_layer_descs(w, h)
    t0 = time.perf_counter()
    c_oracle.run_net(descs, [z[f"w{n}_words"] for n in range(8)], [z[f"b{n}"] for n in range(8)], crop, "dataflow")
    dt = time.perf_counter() - t0
    return {"value": round(w * h / dt / 1e6, 6), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": f"top-left {w}x{h} crop of image 0, all 8 layers, oracle/sicn_oracle.c dataflow form "
                      f"(pad -> sliding-window FSM -> decimate -> folded 8-bit-wrapping MVAU -> bias/ReLU), "
                      f"1 thread, {dt:.1f} s"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from simple_image_compression_network_amd import api

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where the bookkeeping collectives live
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    W, H, B = args.width, args.height, args.images_per_gpu
    # synthetic inputs: uint8 NHWC, i.i.d. uniform 0..255, one seed per global image index
    host = np.stack([np.random.default_rng(rank * B + i).integers(0, 256, (H, W, 3), dtype=np.uint8) for i in range(B)])
    x = torch.from_numpy(host).to(dev)
    # weights: rank 0's PARAM tables, replicated with one broadcast (no-op on one GPU)
    params = api.load_param_weights()
    if world > 1:
        from simple_image_compression_network_amd.dist import broadcast_params
        if rank != 0:
            for pair in params:
                for t in pair:
                    t.m_weights[...] = 0
        broadcast_params(params, src=0, device=cdev)
    net = api.EightLayersNet(W, H, params=params, device=dev)
    out = torch.empty((B,) + net.descs[-1].out_shape, dtype=torch.uint8, device=dev)
    latent = torch.empty((B,) + net.descs[3].out_shape, dtype=torch.uint8, device=dev)
    net.workspace(B)

    def step():
        net.forward(x, out, latent)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    net.profile(True)
    net.layer_ms(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    layer_ms, launches = net.layer_ms(reset=True)
    net.profile(False)

    # bookkeeping collective only: one checksum per rank so rank 0 can report that every shard ran
    chk = torch.tensor([int(out.view(-1)[:: 65537].to(torch.int64).sum().item())], dtype=torch.int64, device=cdev)
    if world > 1:
        chks = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(chks, chk)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    px_per_step = world * B * W * H
    value = px_per_step * args.steps / dt / 1e6
    # ---- roofline of the dominant kernel (largest share of device time) ----------------------
    avg_ms = [m / max(c, 1) for m, c in zip(layer_ms, launches)]
    names = [api._lib.lib().sicn_kernel_for(__import__("ctypes").byref(d.to_c())).decode() for d in net.descs]
    dom = int(np.argmax(avg_ms))
    d = net.descs[dom]
    ops = 2.0 * d.algorithmic_macs * B                     # algorithmic int8 ops per launch (zero-skipped)
    byts = float(B) * (np.prod(d.in_shape) + np.prod(d.out_shape))   # activation bytes per launch
    layers = []
    for l, dd in enumerate(net.descs):
        o = 2.0 * dd.algorithmic_macs * B
        by = float(B) * (np.prod(dd.in_shape) + np.prod(dd.out_shape))
        s = avg_ms[l] * 1e-3
        layers.append({"layer": l, "kernel": names[l], "ms": round(avg_ms[l], 4),
                       "TOPs": round(o / s / 1e12, 2), "GBs": round(by / s / 1e9, 1)})
    traffic = None
    pmc = ROOT / "profiles" / "r01_pmc_summary.json"
    if pmc.exists():
        try:
            traffic = json.loads(pmc.read_text()).get("dominant_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    if names[dom].startswith("mfma"):
        roof = {"bound": "mfma", "achieved": round(ops / (avg_ms[dom] * 1e-3) / 1e12, 2), "peak": PEAK_INT8_TOPS,
                "unit": "TFLOP/s", "frac": round(ops / (avg_ms[dom] * 1e-3) / 1e12 / PEAK_INT8_TOPS, 4)}
    else:
        roof = {"bound": "hbm", "achieved": round(byts / (avg_ms[dom] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": round(byts / (avg_ms[dom] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
    roof.update({"traffic": traffic, "kernel": f"layer {dom} ({names[dom]})", "avg_launch_ms": round(avg_ms[dom], 4),
                 "note": "int8 MAC = 2 ops, counted in the TFLOP/s unit; algorithmic (zero-skipped) work"})

    res = {
        "metric": "Mpixels/s encode+decode (4K RGB)", "value": round(value, 2), "unit": "Mpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic", "config": {"workload": f"{B} x {W}x{H} RGB uint8 images per GPU, eight_layers_net (PARAM weights), "
                                                    f"BASELINE.json configs[3] shard", "images_per_gpu": B,
                                        "global_images": world * B, "parallelism": f"image-sharded x{world}"},
        "roofline": roof, "layers": layers,
        "device_ms_sum_per_step": round(sum(avg_ms), 3),
    }
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(host[0], args.cpu_sample)
    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
