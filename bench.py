#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the 8-layer integer transform (encode L0-L3 + decode L4-L7,
conv_nonsquare_top.cpp:295-357) on synthetic 4K RGB images, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over this rank's batch of IMAGES_PER_GPU 3840x2160x3 uint8
images, already resident in HBM (BASELINE.json configs[3]: 64 images over 8 GPUs = 8 per GPU; weak
scaling: every rank always owns 8 images, no data-path collective — images are independent).
Rank 0 prints ONE JSON line:

  value / ms_per_step   the timed transform (headline, as BASELINE.json's metric)
  output_bit_exact      SHA-256 of every rank's latents and reconstructions of the TIMED run against
                        tests/golden/bench_4k_hashes.json (made by the oracle, tests/golden/make_bench_hashes.py)
  rank_checksums        adler32 of each rank's output batch, gathered on rank 0 (every shard ran, shards differ)
  roofline              the dominant kernel, timed live with hipEvents recorded by the library on the launch
                        stream (sicn_net_profile); `traffic` is the PMC figure of profiles/<round>_pmc_summary.json
                        and is only reported while that file's kernel-source fingerprint matches this build
  with_coder            secondary: analysis -> rANS-W encode -> decode -> synthesis (the coder is this project's own;
                        the reference has none), timed the same way
  cpu_baseline          the oracle's dataflow-faithful C port (1 thread, crop) — the stand-in for the reference's HLS
                        C-simulation, which needs Vivado headers — plus `all_cores`: the oracle's OpenMP direct form
                        on one whole 4K image, whose bytes are also compared with the GPU's image 0, plus `reference`:
                        the reference's own golden convolution (conv.hpp, compiled unmodified into oracle/_ref in the
                        build container; the built library travels, the sources do not) on a 256 x 256 crop
  host_io               secondary: the same step from and to pinned HOST buffers (PCIe-inclusive; never `value`)
  hyperprior            secondary: BASELINE.json configs[4] (GDN / IGDN, hyper stacks, conditional coder)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_INT8_TOPS = 5000.0   # dense int8 MFMA = 2x the ~2.5 PFLOP/s bf16 dense peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0     # HBM3E spec
NOMINAL_SCLK_MHZ = 2400.0  # the shader clock the nominal MFMA peak is quoted at
KERNEL_SOURCES = ("k_common.hpp", "k_l0_common.hpp", "k_mfma16.hip", "k_mfma16p.hip", "k_mfma16x.hip", "k_mfma.hip", "k_rgb.hip", "k_generic.hip", "sicn_plan.h",
                  "sicn_internal.h", "sicn_abi.hip")   # what the eight layers of the headline run through (kernels + every launch decision)


# what the secondary legs (with_coder, hyperprior) run through on top of that: the coders and the activation (the
# profiles/rNN_hyperprior_* evidence is tied to THESE sources)
SECONDARY_SOURCES = ("k_gdn.hip", "k_gdn_body.hpp", "k_l0g.hip", "sicn_codec.hip", "sicn_codec_ctx.inc", "sicn_gdn_internal.h")


def kernel_source_fingerprint(sources=KERNEL_SOURCES) -> str:
    """Identifies the kernel build a PMC summary belongs to (so a stale `traffic` is never reported)."""
    h = hashlib.sha256()
    for name in sources:
        h.update((ROOT / "simple_image_compression_network_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def bench_image_ids(rank: int, world: int, per_gpu: int):
    """Global indices (= seeds = keys of tests/golden/bench_4k_hashes.json) of the images rank `rank` owns in a job of world * per_gpu
    images: the documented partition, image i -> rank i mod world (dist.shard_indices)."""
    from simple_image_compression_network_amd.dist import shard_indices
    return shard_indices(world * per_gpu, rank, world)


class ClockPowerSampler:
    """Shader clock (MHz) and socket power (W) of one GPU, sampled by a side thread through amdsmi while a region runs — never
    inside the stream (`roofline.frac` is against the nominal 5 POP/s = 2.4 GHz; this says what clock the
    chip actually held, so a reader can split the gap into clock and issue without in-kernel stamps)."""

    def __init__(self, pci_bus_id: str | None, period_s: float = 0.004):
        self.period, self.h, self.err = period_s, None, None
        self.regions, self._cur, self._stop, self._thr = {}, None, False, None
        try:
            import amdsmi
            self.smi = amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            pick = None
            for h in hs:
                try:
                    if pci_bus_id and amdsmi.amdsmi_get_gpu_device_bdf(h).lower() == pci_bus_id.lower():
                        pick = h
                except Exception:   # noqa: BLE001
                    pass
            self.h = pick if pick is not None else (hs[0] if len(hs) == 1 else None)
            if self.h is None:
                self.err = f"no amdsmi handle for {pci_bus_id} among {len(hs)}"
        except Exception as e:   # noqa: BLE001
            self.err = f"amdsmi unavailable: {type(e).__name__}: {e}"

    def _read(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        clks = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 60000]
        clk = sum(clks) / len(clks) if clks else m.get("current_gfxclk")
        pw = m.get("current_socket_power")
        if not isinstance(pw, (int, float)) or not 0 < pw < 60000:
            pw = m.get("average_socket_power")
        ok = lambda v: isinstance(v, (int, float)) and 0 < v < 60000
        return (float(clk) if ok(clk) else None, float(pw) if ok(pw) else None)

    def _loop(self):
        import threading  # noqa: F401
        while not self._stop:
            name = self._cur
            if name is not None:
                try:
                    self.regions.setdefault(name, []).append(self._read())
                except Exception as e:   # noqa: BLE001
                    self.err = f"{type(e).__name__}: {e}"
                    return
            time.sleep(self.period)

    def start(self):
        if self.h is None:
            return self
        import threading
        self._thr = threading.Thread(target=self._loop, daemon=True)
        self._thr.start()
        return self

    def region(self, name):
        self._cur = name

    def stop(self):
        self._stop = True
        if self._thr is not None:
            self._thr.join(timeout=1.0)

    def summary(self, name):
        s = self.regions.get(name) or []
        clk = [c for c, _ in s if c is not None]
        pw = [p for _, p in s if p is not None]
        return {"samples": len(s), "sclk_mhz_mean": round(sum(clk) / len(clk), 1) if clk else None,
                "sclk_mhz_min": round(min(clk), 1) if clk else None, "sclk_mhz_max": round(max(clk), 1) if clk else None,
                "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_w_max": round(max(pw), 1) if pw else None}


def device_identity(torch, dev) -> str:
    """hostname + PCI address (+ uuid where torch exposes it) of a rank's GPU: what rank 0 gathers to prove that N DISTINCT
    devices took part in an N-rank run."""
    import socket
    p = torch.cuda.get_device_properties(dev)
    uuid = str(getattr(p, "uuid", "")) if hasattr(p, "uuid") else ""
    if hasattr(p, "pci_bus_id"):
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
    else:   # a torch without the pci_* properties: never a constant (every rank would look like the same device and an nccl run would be
        # refused): the device's index as this process sees it + the visibility mask that maps it to a physical GPU
        idx = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
        bdf = "index%d/visible=%s" % (idx, os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", "all")))
    return f"{socket.gethostname()}|{bdf}|{uuid}"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)   # 40 x 4.2 ms: long enough for the chip to settle at the clock it holds
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--images-per-gpu", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-coder", action="store_true", help="skip the secondary with_coder measurement")
    ap.add_argument("--no-hyperprior", action="store_true", help="skip the secondary hyperprior (configs[4]) measurement")
    ap.add_argument("--no-host-io", action="store_true", help="skip the secondary host-buffer (PCIe-inclusive) measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary small configs (BASELINE.json configs[1], [2])")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s sustained region")
    ap.add_argument("--sustained", action="store_true", help="with --headline-only: keep the >= 2 s sustained region (its clock / power "
                                                             "samples are the steady-state ones; it launches nothing but the 8 layers)")
    ap.add_argument("--headline-only", action="store_true",
                    help="the timed loop and the per-layer table only (what profiles/collect_pmc.sh runs under rocprofv3)")
    ap.add_argument("--cpu-sample", type=int, nargs=2, default=[256, 256], metavar=("W", "H"))
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) is the real thing; gloo lets several ranks share one GPU "
                         "to rehearse the multi-rank path on a 1-GPU box")
    return ap.parse_args()


def cpu_baseline(image0: np.ndarray, sample_wh, gpu_latent0, gpu_out0):
    """Oracle (test infrastructure) used ONLY as the reported CPU baseline and as checker of image 0."""
    from oracle import c_oracle
    from simple_image_compression_network_amd.config import eight_layer_descs
    w, h = sample_wh
    z = np.load(ROOT / "tests" / "golden" / "param_weights.npz")
    words, bias = [z[f"w{n}_words"] for n in range(8)], [z[f"b{n}"] for n in range(8)]
    crop = np.ascontiguousarray(image0[:h, :w])
    t0 = time.perf_counter()
    c_oracle.run_net(eight_layer_descs(w, h), words, bias, crop, "dataflow")
    dt = time.perf_counter() - t0
    res = {"value": round(w * h / dt / 1e6, 6), "unit": "Mpixels/s", "cores": 1, "kind": "port",
           "sample": f"top-left {w}x{h} crop of image 0, all 8 layers, oracle/sicn_oracle.c dataflow form "
                     f"(pad -> sliding-window FSM -> decimate -> folded 8-bit-wrapping MVAU -> bias/ReLU), "
                     f"1 thread, {dt:.1f} s"}
    # all host cores this process may use: OpenMP direct closed form on the whole image 0
    # the GPU box gives one GPU's job a 16-core share of its host (oversubscribing 256 OpenMP threads is slower)
    try:
        nproc = len(os.sched_getaffinity(0))
    except AttributeError:
        nproc = os.cpu_count() or 1
    nproc = max(1, min(nproc, int(os.environ.get("SICN_CPU_THREADS", "16"))))
    H, W = image0.shape[:2]
    t0 = time.perf_counter()
    outs = c_oracle.run_net(eight_layer_descs(W, H), words, bias, image0, "direct", threads=nproc)
    dt = time.perf_counter() - t0
    res["all_cores"] = {"value": round(W * H / dt / 1e6, 4), "unit": "Mpixels/s", "cores": nproc, "nproc": os.cpu_count(),
                        "kind": "port", "sample": f"whole image 0 ({W}x{H}), all 8 layers, oracle direct closed form, "
                                                  f"OpenMP {nproc} threads, gcc -O3, {dt:.1f} s",
                        "gpu_image0_equals_cpu": bool(np.array_equal(outs[3], gpu_latent0) and np.array_equal(outs[7], gpu_out0))}
    # the reference's OWN CPU model where its build travelled with the snapshot: conv_nonsquare<> of the reference's conv.hpp,
    # compiled unmodified in the build container (oracle/_ref, `make -C oracle ref`), all 8 layers of a 256 x 256 crop layer by
    # layer as the reference's testbench runs it (conv3_nonsquare_tb.cpp:861-1056); timed inside the reference code only
    from oracle import ref_conv
    if ref_conv.available():
        from simple_image_compression_network_amd.config import NET_CHANNELS
        crop256 = np.ascontiguousarray(image0[:256, :256])
        routs, rsec = ref_conv.run_net(ref_conv.NET_256, crop256, words, bias, NET_CHANNELS)
        port = c_oracle.run_net(eight_layer_descs(256, 256), words, bias, crop256, "direct", threads=nproc)
        res["reference"] = {"value": round(256 * 256 / rsec / 1e6, 6), "unit": "Mpixels/s", "cores": 1, "kind": "reference",
                            "sample": f"top-left 256x256 crop of image 0, all 8 layers through the reference's conv_nonsquare<> "
                                      f"(conv.hpp:91-123 compiled unmodified, g++ -O2, oracle/_ref), 1 thread, {rsec:.1f} s "
                                      f"inside the reference code",
                            "equals_oracle": bool(all(np.array_equal(a, b) for a, b in zip(routs, port)))}
    else:
        res["reference"] = None   # oracle/_ref is built only where /root/reference exists (build container); see oracle/Makefile
    return res


def small_configs(api, codec, dev):
    """BASELINE.json configs[1] (256 x 256 analysis transform) and configs[2] (1080p encode + decode, without and with the
    coder): one image each, timed by hipGraph replay (a chain of 8 - 12 launches of 10 - 50 us is what these are), outputs
    compared with hashes the oracle made (tests/golden/appendix_a_hashes.json = SURVEY.md Appendix A, config_hashes.json)."""
    import torch
    out = {}
    appx = json.loads((ROOT / "tests" / "golden" / "appendix_a_hashes.json").read_text())
    cfgh = json.loads((ROOT / "tests" / "golden" / "config_hashes.json").read_text())

    def replay_ms(graph, reps):
        for _ in range(10):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def capture(fn):
        import gc
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        gc.collect()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                fn()
        return g

    sha = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()
    # configs[1]: 256 x 256, analysis L0 - L3
    net = api.EightLayersNet(256, 256, device=dev)
    x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1, 256, 256, 3), dtype=np.uint8)).to(dev)
    lat = torch.empty((1,) + net.descs[3].out_shape, dtype=torch.uint8, device=dev)
    g = capture(lambda: net.analysis(x, lat))
    ms = replay_ms(g, 300)
    macs = sum(d.algorithmic_macs for d in net.descs[:4])
    out["256x256_analysis"] = {"ms": round(ms, 4), "Mpixels_per_s": round(256 * 256 / ms / 1e3, 1),
                               "mfma_frac": round(2.0 * macs / (ms * 1e-3) / 1e12 / PEAK_INT8_TOPS, 4),
                               "latent_bit_exact": sha(lat[0]) == appx["layers"]["rng256"][3],
                               "workload": "BASELINE.json configs[1]: one 256x256 RGB tile (seed 0), L0-L3 -> 16x16x192 latent, hipGraph replay"}
    del net, g
    # configs[2]: 1080p, encode + decode
    W, H = 1920, 1080
    net = api.EightLayersNet(W, H, device=dev)
    x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1, H, W, 3), dtype=np.uint8)).to(dev)
    rec = torch.empty((1,) + net.descs[-1].out_shape, dtype=torch.uint8, device=dev)
    lat = torch.empty((1,) + net.descs[3].out_shape, dtype=torch.uint8, device=dev)
    g = capture(lambda: net.forward(x, rec, lat))
    ms = replay_ms(g, 200)
    macs = sum(d.algorithmic_macs for d in net.descs)
    want = cfgh["1080p_seed0"]
    out["1080p_encode_decode"] = {"ms": round(ms, 4), "Mpixels_per_s": round(W * H / ms / 1e3, 1),
                                  "mfma_frac": round(2.0 * macs / (ms * 1e-3) / 1e12 / PEAK_INT8_TOPS, 4),
                                  "bit_exact": sha(lat[0]) == want["latent_sha256"] and sha(rec[0]) == want["recon_sha256"],
                                  "workload": "BASELINE.json configs[2] without the coder: one 1920x1080 RGB image (seed 0), L0-L7, hipGraph replay"}
    lat2 = torch.empty_like(lat)
    # the coder's stream length is an encoder parameter: the default ("auto", codec.auto_stream_symbols) is 8192 for this latent
    # (192 streams, + 2.8 % bytes over the format's 16384); 2048 (765 streams, an eighth of the chain, + 14 % bytes) is the
    # explicit latency-over-rate choice
    for key, ss in (("1080p_with_coder", "auto"), ("1080p_with_coder_short_streams", 2048)):
        coder = codec.LatentCoder(1, *net.descs[3].out_shape, image_width=W, image_height=H, device=dev, stream_symbols=ss)

        def coded():
            net.analysis(x, lat)
            coder.encode(lat)
            coder.decode(lat2)
            net.synthesis(lat2, rec)

        lat2.zero_()
        g2 = capture(coded)
        ms = replay_ms(g2, 200)
        coder.check()
        out[key] = {"ms": round(ms, 4), "Mpixels_per_s": round(W * H / ms / 1e3, 1),
                    "bits_per_pixel": round(8.0 * sum(coder.sizes()) / (W * H), 4), "stream_symbols": coder.stream_symbols,
                    "bit_exact": sha(lat2[0]) == want["latent_sha256"] and sha(rec[0]) == want["recon_sha256"],
                    "workload": "BASELINE.json configs[2]: analysis -> rANS-W encode -> decode -> synthesis, one hipGraph (the coder is this project's own)"}
        del g2, coder
    return out


def main():
    args = parse()
    if args.headline_only:
        args.no_coder = args.no_hyperprior = args.no_host_io = args.no_cpu_baseline = args.no_configs = True
        args.no_sustained = not args.sustained
    import torch
    import torch.distributed as dist

    from simple_image_compression_network_amd import api, codec

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
    # SICN_BENCH_FORCE_DIST=1: create the process group (and run every bookkeeping collective) even for one rank, so that the
    # RCCL code path of the N > 1 runs can be exercised on a 1-GPU box (launch under torch.distributed.run --nproc-per-node 1)
    use_dist = world > 1 or os.environ.get("SICN_BENCH_FORCE_DIST") == "1"
    if world == 1 and use_dist:   # the one-rank rehearsal: dist.py takes its collective paths (the band split's all-gather) with one rank too
        os.environ.setdefault("SICN_FORCE_COLLECTIVES", "1")
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    # which physical device is this rank on?  Gathered on every rank; an N-rank RCCL run on fewer than N distinct devices is refused
    ident = device_identity(torch, dev)
    rank_devices = [ident]
    if use_dist:
        buf = torch.zeros(160, dtype=torch.uint8, device=cdev)
        raw = ident.encode()[:160]
        buf[:len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
        bufs = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(bufs, buf)
        rank_devices = [bytes(b.cpu().tolist()).rstrip(b"\0").decode() for b in bufs]
        if args.backend == "nccl" and len(set(rank_devices)) != world:
            raise SystemExit(f"rank {rank}: {world} ranks on {len(set(rank_devices))} distinct device(s): {rank_devices}")
    sampler = ClockPowerSampler(ident.split("|")[1]).start() if rank == 0 else None

    W, H, B = args.width, args.height, args.images_per_gpu
    # synthetic inputs: uint8 NHWC, i.i.d. uniform 0..255, one seed per global image index
    # partition: the job's world * B images are dealt out as in dist.shard_indices / DESIGN.md section 6 (image i -> rank i mod world)
    image_ids = bench_image_ids(rank, world, B)
    host = np.stack([np.random.default_rng(i).integers(0, 256, (H, W, 3), dtype=np.uint8) for i in image_ids])
    x = torch.from_numpy(host).to(dev)
    # weights: rank 0's PARAM tables, replicated with one broadcast (no-op on one GPU)
    params = api.load_param_weights()
    if use_dist:
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

    class LegFailed(RuntimeError):
        """some rank (not necessarily this one) failed inside a collective section of a secondary leg"""

    def all_ok(ok: bool) -> bool:
        """one MAX all-reduce of a failure flag: every rank learns at the SAME point whether any rank failed"""
        if not use_dist:
            return ok
        f = torch.tensor([0 if ok else 1], dtype=torch.int32, device=cdev)
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        return int(f.item()) == 0

    def guarded(fn):
        """a section of a secondary leg WITHOUT collectives inside: its exception is held until every rank has reported, so
        a failure on one rank only (out of memory, a coder check) never leaves the others blocked in the leg's next collective
        (that used to be a hang, not an error)"""
        err, res = None, None
        try:
            res = fn()
        except Exception as e:   # noqa: BLE001
            err = e
        if not all_ok(err is None):
            raise err if err is not None else LegFailed("another rank failed in this section")
        return res

    def timed(fn, steps, region=None):
        """barrier + synchronize on both sides, MAX over ranks.  An exception inside the loop is held until the closing
        collectives have run on every rank (they stay aligned), then raised everywhere."""
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        if sampler is not None and region:
            sampler.region(region)
        err = None
        t0 = time.perf_counter()
        try:
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            err = e
        if sampler is not None and region:
            sampler.region(None)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if not all_ok(err is None):
            raise err if err is not None else LegFailed("another rank failed in the timed loop")
        return float(t.item())

    secondary_errors = []

    def secondary(name, fn):
        """A secondary measurement must never cost the headline line: an exception is reported in its place, on stderr and in the
        line's top-level `secondary_errors`.  Legs are built from `guarded` sections and `timed` loops, whose failures surface
        on every rank at the same collective, so the ranks stay aligned and go on to the next leg together."""
        try:
            return fn()
        except Exception as e:   # noqa: BLE001 - reported, not swallowed
            import traceback
            traceback.print_exc(file=sys.stderr)
            secondary_errors.append(name)
            return {"error": f"{name}: {type(e).__name__}: {e}"}

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dt = timed(step, args.steps, "timed")   # the headline: nothing but the 8 launches per step on the stream
    # the per-layer table comes from a separate short loop (2 hipEvents per layer and step would sit inside `value` otherwise)
    net.profile(True)
    net.layer_ms(reset=True)
    timed(step, max(10, min(args.steps, 40)), "layers")
    layer_ms, launches = net.layer_ms(reset=True)
    net.profile(False)
    # a sustained region: the same step for >= 2 s (does the rate of the driver-sized region hold once the chip has settled?)
    sustained = None
    if not args.no_sustained:
        ssteps = max(args.steps, int(2.0 / (dt / args.steps)) + 1)
        sdt = timed(step, ssteps, "sustained")
        sustained = {"steps": ssteps, "seconds": round(sdt, 3), "ms_per_step": round(sdt / ssteps * 1e3, 3),
                     "value": round(world * B * W * H * ssteps / sdt / 1e6, 2), "unit": "Mpixels/s"}

    # ---- the TIMED outputs against the committed oracle hashes ------------------------------------
    out_h, lat_h = out.cpu().numpy(), latent.cpu().numpy()
    golden_path = ROOT / "tests" / "golden" / "bench_4k_hashes.json"
    golden = json.loads(golden_path.read_text()) if golden_path.exists() else {}
    verdict = 1   # 1 = all equal, 0 = a mismatch, -1 = no golden entry for some image (or a non-default size)
    for i in range(B):
        g = golden.get(str(image_ids[i])) if (W, H) == (3840, 2160) else None
        if g is None:
            verdict = min(verdict, -1) if verdict != 0 else 0
            continue
        if [hashlib.sha256(lat_h[i].tobytes()).hexdigest(), hashlib.sha256(out_h[i].tobytes()).hexdigest()] != g:
            verdict = 0
    # bookkeeping collectives only: verdict + one checksum per rank, so rank 0 can show every shard ran
    mine = torch.tensor([verdict, zlib.adler32(out_h.reshape(-1)) & 0xFFFFFFFF], dtype=torch.int64, device=cdev)
    gathered = [mine]
    if use_dist:
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
    verdicts = [int(g[0].item()) for g in gathered]
    rank_checksums = [int(g[1].item()) for g in gathered]

    # ---- secondary: the same batch through the entropy coder ---------------------------------------
    def coder_leg():
        lat2 = torch.empty_like(latent)
        coder = guarded(lambda: codec.LatentCoder(B, *net.descs[3].out_shape, image_width=W, image_height=H, device=dev))

        def coder_step():    # four enqueues, no host synchronisation anywhere inside
            net.analysis(x, latent)
            coder.encode(latent)
            coder.decode(lat2)
            net.synthesis(lat2, out)

        guarded(lambda: (coder_step(), coder.check()))
        csteps = max(2, args.steps // 2)
        timed(coder_step, args.warmup)   # untimed warm-up, as for the headline: the leg starts behind seconds of host-side hashing (idle chip)
        cdt = timed(coder_step, csteps)
        guarded(coder.check)   # device-side verdicts of the last step (checksums of the decoded latents included)
        ok = bool(torch.equal(lat2, latent)) and (zlib.adler32(out.cpu().numpy().reshape(-1)) & 0xFFFFFFFF) == rank_checksums[rank]
        return {"value": round(world * B * W * H * csteps / cdt / 1e6, 2), "unit": "Mpixels/s",
                      "ms_per_step": round(cdt / csteps * 1e3, 3), "steps": csteps,
                      "bits_per_pixel": round(8.0 * sum(coder.sizes()) / (B * W * H), 4), "round_trip_exact": ok,
                      "path": "analysis (L0-L3) -> sicn_codec_encode_batch_async (rANS-W) -> sicn_codec_decode_batch_async -> "
                              "synthesis (L4-L7), all enqueued on one stream without host synchronisation",
                      "note": "the coder is this project's own (the reference has none); parity unpinned"}

    with_coder = None if args.no_coder else secondary("with_coder", coder_leg)
    # ---- secondary: BASELINE.json configs[4], the hyperprior configuration (GDN / IGDN main transform, hyper stacks,
    # ---- mode-3 coder for the hyper-latent, mode-4 conditional coder for the latent) on the same batch -----------------
    def hyper_leg():
        from simple_image_compression_network_amd.hyperprior import HyperpriorCodec
        hc = guarded(lambda: HyperpriorCodec(W, H, B, seed=0, device=dev, main_params=params))
        out_h2 = torch.empty_like(out)

        def hyper_step():    # enqueue only: g_a, h_a, z coder, h_s, y coder | z decoder, h_s, y decoder, g_s
            hc.encode(x)
            hc.decode(out_h2)

        guarded(lambda: (hyper_step(), hc.check()))
        hsteps = max(2, args.steps // 4)
        timed(hyper_step, max(1, args.warmup // 2))   # untimed warm-up
        hdt = timed(hyper_step, hsteps)
        guarded(hc.check)
        direct = torch.empty_like(out)
        hc.main.forward(x, direct, want_latent=False)      # the same transform without the coders in between
        # the TIMED run's products against hashes the ORACLE made for the same images (tests/golden/make_hyper_hashes.py through
        # oracle/hyper_pipeline.py): latent, both containers byte for byte, reconstruction (VERDICT r4 item 2 iii: the leg used to be
        # compared with itself only).  Entries exist for images 0 .. 7 at 3840 x 2160; anything else reports None.
        hg_path = ROOT / "tests" / "golden" / "hyper_4k_hashes.json"
        hgold = json.loads(hg_path.read_text()) if hg_path.exists() else {}
        zs_h, ys_h = hc.z_coder.sizes(), hc.y_coder.sizes()
        oracle_ok, checked = None, []
        if (W, H) == (3840, 2160) and hgold.get("gdn_spec_version") == 2:
            sha = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()
            for k, gid in enumerate(image_ids):
                g = hgold.get(str(gid))
                if g is None:
                    continue
                same = (sha(hc.y[k]) == g["y"] and zs_h[k] == g["z_bytes"] and ys_h[k] == g["y_bytes"]
                        and sha(hc.z_coder.slots[k, :zs_h[k]]) == g["z_container"] and sha(hc.y_coder.slots[k, :ys_h[k]]) == g["y_container"]
                        and sha(out_h2[k]) == g["recon"])
                checked.append(gid)
                oracle_ok = same if oracle_ok is None else (oracle_ok and same)
        # two jobs in flight: the ENCODE of batch k + 1 on one stream beside the DECODE of batch k on another (two codec objects, events for
        # the hand-over) — a service that both encodes and decodes.  The coders are latency-bound chains on a few waves per SIMD; beside a
        # transform kernel of the other job they cost little.  Reported NEXT to the one-stream number, which stays `ms_per_step`.
        two = None
        try:   # built from `guarded` sections and `timed` loops like every leg: a failure surfaces on every rank at the same collective
            hc2 = guarded(lambda: HyperpriorCodec(W, H, B, seed=0, device=dev, main_params=params))
            pair, outs2 = (hc, hc2), (out_h2, guarded(lambda: torch.empty_like(out)))
            se, sd = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            state = {"i": 0, "enc": [None, None], "dec": [None, None]}

            def two_step():
                i = state["i"]
                c, o = pair[i & 1], outs2[i & 1]
                with torch.cuda.stream(se):
                    if state["dec"][i & 1] is not None:
                        se.wait_event(state["dec"][i & 1])    # this object's buffers are free again
                    c.encode(x)
                    state["enc"][i & 1] = se.record_event()
                with torch.cuda.stream(sd):
                    sd.wait_event(state["enc"][i & 1])
                    c.decode(o)
                    state["dec"][i & 1] = sd.record_event()
                state["i"] = i + 1

            se.wait_stream(torch.cuda.current_stream())
            sd.wait_stream(torch.cuda.current_stream())
            nsteps2 = 2 * (hsteps // 2 + 1)
            timed(two_step, 2)
            tdt = timed(two_step, nsteps2)
            guarded(lambda: (hc.check(), hc2.check()))
            two = {"ms_per_step": round(tdt / nsteps2 * 1e3, 3), "value": round(world * B * W * H * nsteps2 / tdt / 1e6, 2), "unit": "Mpixels/s",
                   "steps": nsteps2, "outputs_equal_one_stream": bool(torch.equal(outs2[0], direct)) and bool(torch.equal(outs2[1], direct)),
                   "what": "encode(batch k + 1) on one stream beside decode(batch k) on another: two HyperpriorCodec objects, events for the hand-over"}
            del hc2
        except Exception as e:   # noqa: BLE001   (a secondary of a secondary: reported in its place, never takes the leg down; every rank
            two = {"error": f"{type(e).__name__}: {e}"}   # arrives here together because the sections above are rank-aligned)
        return {"containers_and_outputs_equal_oracle": oracle_ok, "oracle_checked_images": checked, "two_jobs_in_flight": two,
                "oracle_check": "sha256 of latent, z container, y container and reconstruction of the timed run vs tests/golden/hyper_4k_hashes.json "
                                "(oracle/hyper_pipeline.py; GDN specification version 2)",
                "value": round(world * B * W * H * hsteps / hdt / 1e6, 2), "unit": "Mpixels/s",
                 "ms_per_step": round(hdt / hsteps * 1e3, 3), "steps": hsteps,
                 "bits_per_pixel": round(8.0 * sum(hc.bytes_per_image()) / (B * W * H), 4),
                 "round_trip_exact": bool(torch.equal(hc.y_hat, hc.y)) and bool(torch.equal(out_h2, direct)),
                 "path": "g_a (L0-L3, GDN) -> h_a -> rANS-W(z) -> h_s -> rANS-WC(y | scale map, checkerboard context) | decode: "
                         "rANS-W(z) -> h_s -> rANS-WC(y) -> g_s (L4-L7, IGDN); one stream, no host synchronisation",
                 "note": "BASELINE.json configs[4]; no reference counterpart (SURVEY.md section 0): parity unpinned, seeded random "
                         "hyper / GDN parameters"}

    hyper = None if args.no_hyperprior else secondary("hyperprior", hyper_leg)
    # ---- secondary: the same step with HOST buffers on both sides (pinned memory, upload / compute / download streams) ----
    def host_io_leg():
        from simple_image_compression_network_amd.host_pipeline import HostPipeline
        hp = HostPipeline(net, B, depth=2, want_latent=True)
        nb = 3                                               # distinct host batches in flight (the data repeats: synthetic)
        h_in = [HostPipeline.pinned(x.shape) for _ in range(nb)]
        h_out = [HostPipeline.pinned(out.shape) for _ in range(nb)]
        h_lat = [HostPipeline.pinned(latent.shape) for _ in range(nb)]
        for t in h_in:
            t.copy_(x)
        torch.cuda.synchronize()
        psteps = max(3, args.steps)
        idx = [i % nb for i in range(psteps)]
        hp.run([h_in[i] for i in idx[:nb]], [h_out[i] for i in idx[:nb]], [h_lat[i] for i in idx[:nb]])   # warm-up
        hp.synchronize()
        t0 = time.perf_counter()
        hp.run([h_in[i] for i in idx], [h_out[i] for i in idx], [h_lat[i] for i in idx])
        hp.synchronize()
        pdt = time.perf_counter() - t0
        same = all((zlib.adler32(t.numpy().reshape(-1)) & 0xFFFFFFFF) == rank_checksums[rank] for t in h_out)
        step_bytes = x.numel() + out.numel() + latent.numel()
        return {"value": round(B * W * H * psteps / pdt / 1e6, 2), "unit": "Mpixels/s", "ms_per_step": round(pdt / psteps * 1e3, 3),
                "steps": psteps, "host_bytes_per_step": int(step_bytes),
                "pcie_GBs_each_way": round(max(x.numel(), out.numel() + latent.numel()) * psteps / pdt / 1e9, 1),
                "outputs_equal_resident_run": bool(same),
                "path": "pinned host batch -> H2D stream -> eight_layers_net on the compute stream -> D2H stream (reconstruction + "
                        "latent), 2 device slots, events between the three streams (host_pipeline.HostPipeline)",
                "note": "never `value`: the headline is HBM-resident; this is the rate a caller that owns host streams sees"}

    pcie = secondary("host_io", host_io_leg) if (not args.no_host_io and rank == 0 and world == 1) else None
    # ---- secondaries of the multi-GPU runs (N > 1): strong scaling and the single-image band split ----------------------
    def strong_leg():
        per = max(1, 64 // world)                      # BASELINE.json configs[3]: 64 images in all, 64 / N per rank

        def setup():
            xs = x[:per] if per <= B else torch.cat([x] * ((per + B - 1) // B))[:per]
            outs_s = torch.empty((per,) + net.descs[-1].out_shape, dtype=torch.uint8, device=dev)
            lat_s = torch.empty((per,) + net.descs[3].out_shape, dtype=torch.uint8, device=dev)
            net.forward(xs, outs_s, lat_s)
            return xs, outs_s, lat_s

        xs, outs_s, lat_s = guarded(setup)
        ssteps = max(4, args.steps // 2)
        sdt = timed(lambda: net.forward(xs, outs_s, lat_s), ssteps)
        return {"value": round(world * per * W * H * ssteps / sdt / 1e6, 2), "unit": "Mpixels/s", "scaling": "strong",
                "images_per_gpu": per, "global_images": world * per, "ms_per_step": round(sdt / ssteps * 1e3, 3)}

    def banded_leg():
        # one image over the N ranks by horizontal bands (device-resident with nccl: one all-gather of the kept rows over xGMI)
        from simple_image_compression_network_amd import dist as sdist
        img = x[:1].clone()
        if args.backend == "nccl":
            dist.broadcast(img, src=0)
        else:
            t = img.cpu()
            dist.broadcast(t, src=0)
            img = t.to(dev)
        bnet = sdist.BandedNet(W, H, world, rank, params=params, device=dev)
        rec_b = bnet.forward(img)
        bsteps = max(4, args.steps // 2)
        bdt = timed(lambda: bnet.forward(img), bsteps)
        whole = torch.empty((1,) + net.descs[-1].out_shape, dtype=torch.uint8, device=dev)
        net.forward(img, whole, want_latent=False)
        torch.cuda.synchronize()
        return {"ms_per_image": round(bdt / bsteps * 1e3, 3), "value": round(W * H * bsteps / bdt / 1e6, 2), "unit": "Mpixels/s",
                "bands": world, "equals_one_gpu_bytes": bool(torch.equal(rec_b, whole)),
                "note": "latency of ONE image: each rank computes a band with 64 rows of recomputed halo, one all-gather of the kept rows"}

    strong = banded = None
    if (world > 1 or os.environ.get("SICN_BENCH_FORCE_DIST") == "1") and not args.headline_only:
        strong = secondary("strong_scaling", strong_leg)
        # the band split has a data-path collective (the all-gather of the kept rows) INSIDE its step: a rank that fails there
        # cannot be waited for; it runs last of the collective legs so that nothing else is lost with it
        banded = secondary("banded", banded_leg)
    if sampler is not None:
        sampler.stop()
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    import ctypes
    px_per_step = world * B * W * H
    value = px_per_step * args.steps / dt / 1e6
    # ---- roofline of the dominant kernel (largest share of device time) ----------------------
    avg_ms = [m / max(c, 1) for m, c in zip(layer_ms, launches)]
    names = [api._lib.lib().sicn_kernel_for(ctypes.byref(d.to_c())).decode() for d in net.descs]
    dom = int(np.argmax(avg_ms))
    d = net.descs[dom]
    ops = 2.0 * d.algorithmic_macs * B                     # algorithmic int8 ops per launch (zero-skipped)
    byts = float(B) * (np.prod(d.in_shape) + np.prod(d.out_shape))   # algorithmic activation bytes per launch
    layers = []
    for l, dd in enumerate(net.descs):
        o = 2.0 * dd.algorithmic_macs * B
        by = float(B) * (np.prod(dd.in_shape) + np.prod(dd.out_shape))
        s = avg_ms[l] * 1e-3
        layers.append({"layer": l, "kernel": names[l], "ms": round(avg_ms[l], 4),
                       "TOPs": round(o / s / 1e12, 2), "GBs": round(by / s / 1e9, 1)})
    # HBM traffic of the dominant kernel: PMC counters cannot be collected inside this run; the figure comes from the
    # newest profiles/*_pmc_summary.json and only if it was measured on exactly these kernel sources
    traffic, traffic_source = None, None
    fp = kernel_source_fingerprint()
    for pmc in sorted((ROOT / "profiles").glob("r*_pmc_summary.json"), reverse=True):
        try:
            j = json.loads(pmc.read_text())
        except Exception:
            continue
        if j.get("kernel_source_fingerprint") == fp and j.get("dominant_layer") == dom and (W, H, B) == (3840, 2160, 8):
            traffic = j.get("dominant_kernel_hbm_bytes_per_launch")
            traffic_source = f"profiles/{pmc.name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same kernel sources {fp})"
            break
    if traffic is None:
        traffic_source = f"no PMC summary for kernel sources {fp}: rerun profiles/collect_pmc.sh"
    if names[dom].startswith("mfma"):
        roof = {"bound": "mfma", "achieved": round(ops / (avg_ms[dom] * 1e-3) / 1e12, 2), "peak": PEAK_INT8_TOPS,
                "unit": "TFLOP/s", "frac": round(ops / (avg_ms[dom] * 1e-3) / 1e12 / PEAK_INT8_TOPS, 4)}
    else:
        roof = {"bound": "hbm", "achieved": round(byts / (avg_ms[dom] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": round(byts / (avg_ms[dom] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
    # the clock and power the chip held (side-thread amdsmi samples): frac stays against the nominal peak; frac_at_clock =
    # achieved / (peak x sclk / 2400 MHz) is the issue efficiency at the clock actually held.  The gpu_metrics table the samples
    # come from is smoothed over a window that is LONGER than the 75 ms regions (a region that starts from idle reads 1.6 - 1.9 GHz
    # and 0.7 - 1.0 kW while running as fast, to 0.5 %, as the >= 2 s region that reads 2.09 GHz and 1.34 kW), so the steady
    # state of the sustained region is the figure used; the short regions' readings are kept in `clocks` as they came
    clk_l, clk_t, clk_s = sampler.summary("layers"), sampler.summary("timed"), sampler.summary("sustained")
    clk = clk_s if clk_s["sclk_mhz_mean"] else clk_l
    roof.update({"sclk_mhz_mean": clk["sclk_mhz_mean"], "power_w_mean": clk["power_w_mean"],
                 "frac_at_clock": (round(roof["frac"] * NOMINAL_SCLK_MHZ / clk["sclk_mhz_mean"], 4)
                                   if clk["sclk_mhz_mean"] and roof["bound"] == "mfma" else None),
                 "clock_source": sampler.err or f"amdsmi gpu_metrics (mean of the XCDs' current_gfxclk, current_socket_power), "
                                                f"{clk['samples']} samples at {int(sampler.period * 1e3)} ms over the "
                                                f"{'sustained (>= 2 s)' if clk is clk_s else 'per-layer'} region; "
                                                f"nominal peak assumes {NOMINAL_SCLK_MHZ} MHz"})
    roof.update({"traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes": int(byts),
                 "kernel": f"layer {dom} ({names[dom]})", "avg_launch_ms": round(avg_ms[dom], 4),
                 "note": "int8 MAC = 2 ops, counted in the TFLOP/s unit; algorithmic (zero-skipped) work"})
    net_ops = sum(2.0 * dd.algorithmic_macs for dd in net.descs) * B
    res = {
        "metric": "Mpixels/s encode+decode (4K RGB)", "value": round(value, 2), "unit": "Mpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic", "config": {"workload": f"{B} x {W}x{H} RGB uint8 images per GPU, eight_layers_net (PARAM weights), "
                                                    f"BASELINE.json configs[3] shard; transform only (the reference has no coder)",
                                        "images_per_gpu": B, "global_images": world * B, "parallelism": f"image-sharded x{world}",
                                        "partition": "image i -> rank i mod world (dist.shard_indices)", "image_ids_rank0": image_ids},
        "output_bit_exact": (None if any(v < 0 for v in verdicts) and all(v != 0 for v in verdicts) else all(v == 1 for v in verdicts)),
        "output_check": "sha256 of every latent and reconstruction of the timed run vs tests/golden/bench_4k_hashes.json (oracle direct form)",
        "rank_checksums": rank_checksums,
        "rank_devices": rank_devices, "distinct_devices": len(set(rank_devices)),
        "world_size": (dist.get_world_size() if use_dist else 1),
        "clocks": {"timed": clk_t, "layers": clk_l, "sustained": clk_s},
        "roofline": roof, "layers": layers,
        "kernel_source_fingerprint": fp, "secondary_source_fingerprint": kernel_source_fingerprint(SECONDARY_SOURCES),
        "device_ms_sum_per_step": round(sum(avg_ms), 3),
        "whole_net_mfma_frac": round(net_ops / (dt / args.steps) / 1e12 / PEAK_INT8_TOPS, 4),
        # SURVEY.md 8(d), the memory view: layer-wise algorithmic bytes (every activation written once and read once)
        "whole_net_hbm_frac": round(sum(l["GBs"] * l["ms"] * 1e-3 for l in layers) / (dt / args.steps) / PEAK_HBM_GBS, 4)
        if all("GBs" in l for l in layers) else None,
    }
    if sustained is not None:
        res["sustained"] = sustained
    if strong is not None:
        res["strong_scaling"] = strong
    if banded is not None:
        res["banded"] = banded
    if not args.no_configs and world == 1:
        res["configs"] = secondary("configs", lambda: small_configs(api, codec, dev))
    if pcie is not None:
        res["host_io"] = pcie
    if with_coder is not None:
        res["with_coder"] = with_coder
    if hyper is not None:
        res["hyperprior"] = hyper
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = secondary("cpu_baseline", lambda: cpu_baseline(host[0], args.cpu_sample, lat_h[0], out_h[0]))
    res["secondary_errors"] = secondary_errors
    if use_dist:
        res["config"]["collectives"] = f"torch.distributed backend {dist.get_backend()} (bookkeeping only: weight broadcast, barrier, MAX of the timed region, checksum all-gather)"
    print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
