#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect_pmc.sh into the tracked evidence:
  profiles/<tag>_kernel_stats.csv   copy of the --stats kernel summary of `python3 bench.py`
  profiles/<tag>_pmc_summary.json   per-kernel HBM bytes (FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md
                                    + WRITE_SIZE), L2 hit rate, MFMA pipe utilisation, wait fractions; the dominant
                                    kernel's bytes per launch next to its algorithmic bytes; the fingerprint of the
                                    kernel sources (bench.py only reports `traffic` while it matches)
usage: summarize_pmc.py <tag> <gpurun_out/<tag>_prof>"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
tag, out = sys.argv[1], Path(sys.argv[2])


def short(n):
    return n.replace("void sicn::", "").replace("sicn::", "").replace("xw::", "").split("(")[0]


def is_layer_kernel(name):
    return name.startswith(("k_l0", "k_l7", "k_conv", "k_deconv", "k_mfma", "k_generic"))


def by_layer(rows, key):
    """The headline-only bench launches nothing but the net's 8 layers, in order, step after step: dispatch i (sorted by `key`)
    of the layer kernels is layer i mod 8.  (Kernel name + grid no longer identifies a layer: the wide persistent kernels run
    layers 1 / 2 and 5 / 6 with the same 256 workgroups.)  Checked: every layer sees one kernel name only."""
    rows = sorted((r for r in rows if is_layer_kernel(short(r["Kernel_Name"]))), key=key)
    out = collections.defaultdict(list)
    for i, r in enumerate(rows):
        out[i % 8].append(r)
    for l, rs in out.items():
        names = {short(r["Kernel_Name"]) for r in rs}
        if len(names) != 1:
            raise SystemExit(f"layer {l}: dispatch order does not repeat with period 8: {names}")
    if not out[0] or not short(out[0][0]["Kernel_Name"]).startswith("k_l0") or not short(out[7][0]["Kernel_Name"]).startswith("k_l7"):
        raise SystemExit("the dispatch sequence does not start with layer 0 / end with layer 7: " + str({l: short(rs[0]["Kernel_Name"]) for l, rs in out.items()}))
    return out


def counters(d):
    rows = []
    for f in glob.glob(str(d) + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows:
        return {}
    # one row per (dispatch, counter): order the dispatches first
    disp = {}
    for r in rows:
        disp.setdefault(int(r["Dispatch_Id"]), r)
    layers = by_layer(disp.values(), key=lambda r: int(r["Dispatch_Id"]))
    layer_of = {int(r["Dispatch_Id"]): l for l, rs in layers.items() for r in rs}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    names = {}
    for r in rows:
        l = layer_of.get(int(r["Dispatch_Id"]))
        if l is None:
            continue
        names[l] = short(r["Kernel_Name"])
        acc[l][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {f"layer {l} {names[l]}": {c: sum(v) / len(v) for c, v in cs.items()} for l, cs in acc.items()}


stats = glob.glob(str(out / "stats") + "/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], ROOT / "profiles" / f"{tag}_kernel_stats.csv")
# per LAYER: this is the table the bench line's roofline.avg_launch_ms has to agree with
per = {}
trace = []
for f in glob.glob(str(out / "stats") + "/**/*kernel_trace.csv", recursive=True):
    trace += list(csv.DictReader(open(f)))
if trace:
    layers = by_layer(trace, key=lambda r: int(r["Start_Timestamp"]))
    with open(ROOT / "profiles" / f"{tag}_per_layer_dispatch.csv", "w") as fh:
        fh.write("layer,kernel,grid_threads,calls,avg_ns,min_ns,max_ns\n")
        for l in sorted(layers):
            v = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in layers[l]]
            r0 = layers[l][0]
            per[l] = v
            fh.write(f'{l},"{short(r0["Kernel_Name"])}",{r0.get("Grid_Size", r0.get("Grid_Size_X", ""))},{len(v)},{sum(v) / len(v):.0f},{min(v)},{max(v)}\n')
fetch, write, sq = counters(out / "pmc_fetch"), counters(out / "pmc_write"), counters(out / "pmc_sq")
kernels = {}
for k in sorted(set(fetch) | set(write) | set(sq)):
    f, w, s = fetch.get(k, {}), write.get(k, {}), sq.get(k, {})
    e = {}
    if "FETCH_SIZE" in f:
        e["fetch_bytes_reported"] = f["FETCH_SIZE"] * 1024
        e["fetch_bytes_x2_gfx950_correction"] = 2 * f["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in w:
        e["write_bytes"] = w["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in f and "TCC_MISS_sum" in w and f["TCC_HIT_sum"] + w["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = round(f["TCC_HIT_sum"] / (f["TCC_HIT_sum"] + w["TCC_MISS_sum"]), 3)
    if "GRBM_GUI_ACTIVE" in s and s["GRBM_GUI_ACTIVE"] > 0:
        if "SQ_VALU_MFMA_BUSY_CYCLES" in s:
            e["mfma_pipe_util"] = round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (s["GRBM_GUI_ACTIVE"] / 8), 3)
        e["grbm_gui_active"] = s["GRBM_GUI_ACTIVE"]
    if s.get("SQ_WAVE_CYCLES", 0) > 0:
        e["wait_any_frac"] = round(s.get("SQ_WAIT_ANY", 0) / s["SQ_WAVE_CYCLES"], 3)
        e["wait_inst_frac"] = round(s.get("SQ_WAIT_INST_ANY", 0) / s["SQ_WAVE_CYCLES"], 3)
    if s.get("SQ_LDS_ACTIVE", 0) > 0:
        e["lds_bank_conflict_frac"] = round(s.get("SQ_LDS_BANK_CONFLICT", 0) / s["SQ_LDS_ACTIVE"], 3)
        if s.get("GRBM_GUI_ACTIVE", 0) > 0:   # SQ_LDS_ACTIVE sums the CUs' LDS-busy cycles: share of the launch an average CU's LDS was busy
            e["lds_active_frac"] = round(s["SQ_LDS_ACTIVE"] / 256 / (s["GRBM_GUI_ACTIVE"] / 8), 3)
    kernels[k] = e

# dominant layer of the 8 x 4K bench: the largest average launch in the stats run
from bench import kernel_source_fingerprint  # noqa: E402
summary = {"_about": "rocprofv3 --pmc passes of `python3 bench.py --headline-only --steps 3 --warmup 1` (8 x 4K images per "
                     "launch; nothing but the 8 layers runs, so dispatch i is layer i mod 8). Separate passes {FETCH_SIZE,TCC_HIT_sum}, {WRITE_SIZE,TCC_MISS_sum}, {SQ_*,GRBM_GUI_ACTIVE}. "
                     "FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide "
                     "streaming read, so the x2 figure is the one to compare with byte counts (MI355X_MICROARCH.md, HBM). "
                     "mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs). Key = layer + kernel.",
           "kernel_source_fingerprint": kernel_source_fingerprint(), "kernels": kernels}
try:
    line = [l for l in open(out / "bench_under_rocprof.json") if l.startswith("{")][-1]
    b = json.loads(line)
    dom = int(b["roofline"]["kernel"].split()[1])
    summary["dominant_layer"] = dom
    summary["dominant_kernel_algorithmic_bytes_per_launch"] = b["roofline"]["algorithmic_bytes"]
    summary["dominant_kernel_avg_launch_ms_bench"] = b["roofline"]["avg_launch_ms"]
    summary["bench_under_rocprof"] = {k: b[k] for k in ("value", "ms_per_step", "layers", "output_bit_exact")}
    # the clock and power the chip held while these numbers were taken (bench.py's side-thread amdsmi sampler)
    summary["sclk_mhz_mean"] = b["roofline"].get("sclk_mhz_mean")
    summary["power_w_mean"] = b["roofline"].get("power_w_mean")
    summary["clocks"] = b.get("clocks")
    best = next((k for k in kernels if k.startswith(f"layer {dom} ")), None)
    if best:
        e = kernels[best]
        summary["dominant_kernel"] = best
        if "fetch_bytes_x2_gfx950_correction" in e and "write_bytes" in e:
            summary["dominant_kernel_hbm_bytes_per_launch"] = int(e["fetch_bytes_x2_gfx950_correction"] + e["write_bytes"])
        if dom in per:
            v = per[dom]
            summary["dominant_kernel_avg_launch_ms_rocprof"] = round(sum(v) / len(v) / 1e6, 4)
            ops = 2.0 * 8 * 518400 * 128 * 128 * 25 if dom in (1, 6) else None
            if ops:
                summary["dominant_kernel_roofline_frac_from_rocprof"] = round(ops / (sum(v) / len(v) * 1e-9) / 5e15, 4)
    summary["per_layer_avg_ms_rocprof"] = {str(l): round(sum(v) / len(v) / 1e6, 4) for l, v in sorted(per.items())}
except Exception as ex:  # noqa: BLE001
    summary["_warning"] = f"bench line not parsed: {ex}"
(ROOT / "profiles" / f"{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1) + "\n")
print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}, indent=1))
