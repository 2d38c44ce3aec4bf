#!/usr/bin/env python3
"""Per-kernel PMC summary of tools/hyper_pmc.sh: for every kernel name + grid size (k_gdn runs on tensors of three sizes) the mean over its
dispatches of: duration, VALU busy (4 x SQ_ACTIVE_INST_VALU — the counter is in quad-cycles — / 1024 SIMDs / cycles of the dispatch: the share
of cycles a SIMD's vector issue is taken, MFMA issue included),
MFMA pipe busy, the wave-cycle split (waiting at s_waitcnt / barrier, issue stalls, issuing) and HBM bytes (FETCH_SIZE x 2 per the
gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE; both reported in KiB — the conversion of profiles/summarize_pmc.py).
usage: hyper_pmc_summary.py <dir>"""
import collections
import csv
import glob
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "profiles"))
d = Path(sys.argv[1])


def short(n):
    return n.replace("void sicn::", "").replace("sicn::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def rows(sub):
    out = []
    for f in glob.glob(str(d / sub) + "/**/*counter_collection.csv", recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def trace(sub):
    out = {}
    for f in glob.glob(str(d / sub) + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return out


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "fetch", "write"):
    dur = trace(sub)
    seen = set()
    for r in rows(sub):
        key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        did = int(r["Dispatch_Id"])
        if sub == "sq" and did not in seen and did in dur:
            seen.add(did)
            acc[key]["us"].append(dur[did])
N_CU, SIMDS = 256, 4
print(f"{'kernel':44s} {'grid':>9s} {'n':>3s} {'us':>8s} {'VALU busy':>9s} {'MFMA busy':>9s} {'wait':>6s} {'stall':>6s} {'issue':>6s} {'HBM r MB':>9s} {'HBM w MB':>9s} {'TB/s':>6s}")
for (name, grid), c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("us", [0]))):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    if "us" not in m or m["us"] < 20:
        continue
    cu_cycles = m.get("GRBM_GUI_ACTIVE", 0) / 8       # the counter sums the 8 XCDs (profiles/summarize_pmc.py): cycles of the dispatch
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    valu = 4 * m.get("SQ_ACTIVE_INST_VALU", 0) / (N_CU * SIMDS) / cu_cycles if cu_cycles else 0    # SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)
    mfma = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (N_CU * SIMDS) / cu_cycles if cu_cycles else 0
    rd, wr = 2 * m.get("FETCH_SIZE", 0) * 1024, m.get("WRITE_SIZE", 0) * 1024     # KB units -> bytes; FETCH_SIZE x 2 on gfx950
    print(f"{name[:44]:44s} {grid:9d} {len(c['us']):3d} {m['us']:8.1f} {valu:9.3f} {mfma:9.3f} {m.get('SQ_WAIT_ANY', 0) / wc:6.2f} "
          f"{m.get('SQ_WAIT_INST_ANY', 0) / wc:6.2f} {m.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.2f} {rd / 1e6:9.1f} {wr / 1e6:9.1f} {(rd + wr) / m['us'] / 1e6:6.2f}")
