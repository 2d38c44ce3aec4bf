#!/bin/bash
# Instruction-cache counters of the headline run (bench.py --headline-only: nothing but the 8 layers): the wide persistent kernels' tile loop is
# 87 - 103 KB of code against a 64 KB instruction cache shared by two CUs.  One --pmc pass (kernel trace only).  usage: bash tools/icache_pmc.sh <tag>
tag=${1:-icache}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd "$root" && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$out" -o pmc --output-format csv -- python3 bench.py --headline-only --steps 3 --warmup 1 > /dev/null 2> "$out/err.txt"
python3 - "$out" <<'PY' | tee $root/gpurun_out/${tag}_icache.txt
import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
disp = {}
for r in rows:
    disp.setdefault(int(r["Dispatch_Id"]), r["Kernel_Name"])
layer_k = [i for i in sorted(disp) if any(s in disp[i] for s in ("k_l0", "k_l7", "k_conv", "k_deconv"))]
layer = {i: n % 8 for n, i in enumerate(layer_k)}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
name = {}
for r in rows:
    l = layer.get(int(r["Dispatch_Id"]))
    if l is None:
        continue
    name[l] = r["Kernel_Name"].replace("void sicn::", "").replace("sicn::", "").split("(")[0]
    acc[l][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'layer':28s} {'icache req':>12s} {'hits':>12s} {'misses':>12s} {'dup':>10s} {'miss rate':>9s} {'ifetch level / wave-cycle':>26s} {'cycles':>10s} {'misses per 1k cycles per CU pair':>32s}")
for l in range(8):
    m = {k: sum(v) / len(v) for k, v in acc[l].items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    req = m.get("SQC_ICACHE_REQ", 0) or 1
    print(f"{l} {name.get(l, '?')[:26]:26s} {m.get('SQC_ICACHE_REQ', 0):12.0f} {m.get('SQC_ICACHE_HITS', 0):12.0f} {m.get('SQC_ICACHE_MISSES', 0):12.0f} "
          f"{m.get('SQC_ICACHE_MISSES_DUPLICATE', 0):10.0f} {m.get('SQC_ICACHE_MISSES', 0) / req:9.3f} {m.get('SQ_IFETCH_LEVEL', 0) / (m.get('SQ_WAVE_CYCLES', 0) or 1):26.3f} "
          f"{cyc:10.0f} {m.get('SQC_ICACHE_MISSES', 0) / 128 / (cyc / 1000 if cyc else 1):32.2f}")
PY
