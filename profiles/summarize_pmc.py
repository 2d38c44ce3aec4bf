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
    return n.replace("void sicn::", "").replace("sicn::", "").split("(")[0]


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(str(d) + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = short(r["Kernel_Name"])
            if "k_" not in name:
                continue
            key = f'{name} {r.get("Grid_Size", r.get("Grid_Size_X", ""))}'
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


stats = glob.glob(str(out / "stats") + "/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], ROOT / "profiles" / f"{tag}_kernel_stats.csv")
# the same trace per (kernel, grid size): one kernel serves several layers (layer 5 and 6 are the same deconv kernel), and the
# bench line's roofline.avg_launch_ms is per LAYER — this is the table it has to agree with
per = collections.defaultdict(list)
for f in glob.glob(str(out / "stats") + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = short(r["Kernel_Name"])
        grid = r.get("Grid_Size", r.get("Grid_Size_X", ""))
        per[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
if per:
    with open(ROOT / "profiles" / f"{tag}_per_layer_dispatch.csv", "w") as fh:
        fh.write("kernel,grid_threads,calls,avg_ns,min_ns,max_ns\n")
        for (name, grid), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            fh.write(f'"{name}",{grid},{len(v)},{sum(v) / len(v):.0f},{min(v)},{max(v)}\n')
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
    kernels[k] = e

# dominant layer of the 8 x 4K bench: the largest average launch in the stats run
from bench import kernel_source_fingerprint  # noqa: E402
summary = {"_about": "rocprofv3 --pmc passes of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-coder` (8 x 4K images per "
                     "launch). Separate passes {FETCH_SIZE,TCC_HIT_sum}, {WRITE_SIZE,TCC_MISS_sum}, {SQ_*,GRBM_GUI_ACTIVE}. "
                     "FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide "
                     "streaming read, so the x2 figure is the one to compare with byte counts (MI355X_MICROARCH.md, HBM). "
                     "mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs). Key = kernel + grid size.",
           "kernel_source_fingerprint": kernel_source_fingerprint(), "kernels": kernels}
try:
    line = [l for l in open(out / "bench_under_rocprof.json") if l.startswith("{")][-1]
    b = json.loads(line)
    dom = int(b["roofline"]["kernel"].split()[1])
    summary["dominant_layer"] = dom
    summary["dominant_kernel_algorithmic_bytes_per_launch"] = b["roofline"]["algorithmic_bytes"]
    summary["dominant_kernel_avg_launch_ms_bench"] = b["roofline"]["avg_launch_ms"]
    summary["bench_under_rocprof"] = {k: b[k] for k in ("value", "ms_per_step", "layers", "output_bit_exact")}
    # the dominant layer's kernel: the MFMA conv/deconv kernel with the largest grid among its family
    fam = "true" if b["layers"][dom]["kernel"] == "mfma_deconv" else "false"
    kind = b["layers"][dom]["kernel"]
    if kind.startswith("mfma"):
        pre = "k_deconv_p" if kind == "mfma_deconv" else "k_conv_p"
        cands = [k for k in kernels if k.startswith(pre)] or [k for k in kernels if k.startswith("k_mfma16_t") and f", {fam}," in k]
    else:
        cands = [k for k in kernels if k.startswith("k_l0" if dom == 0 else "k_l7")]
    if cands:
        best = max(cands, key=lambda k: int(k.split()[-1] or 0))
        e = kernels[best]
        summary["dominant_kernel"] = best
        if "fetch_bytes_x2_gfx950_correction" in e and "write_bytes" in e:
            summary["dominant_kernel_hbm_bytes_per_launch"] = int(e["fetch_bytes_x2_gfx950_correction"] + e["write_bytes"])
        name, grid = best.rsplit(" ", 1)
        if (name, grid) in per:
            v = per[(name, grid)]
            summary["dominant_kernel_avg_launch_ms_rocprof"] = round(sum(v) / len(v) / 1e6, 4)
except Exception as ex:  # noqa: BLE001
    summary["_warning"] = f"bench line not parsed: {ex}"
(ROOT / "profiles" / f"{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1) + "\n")
print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}, indent=1))
