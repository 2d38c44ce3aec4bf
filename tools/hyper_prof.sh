#!/bin/bash
# Six hyperprior steps (8 x 4K) under rocprofv3: per-kernel averages -> gpurun_out/<tag>_hyper_kernels.txt
# usage (on the GPU box, from the repo root): bash tools/hyper_prof.sh <tag> [options as k=v,k=v]
tag=${1:-hyper}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/tools/hyper_once.py "$2" > $root/gpurun_out/${tag}_hyper_once.txt 2>&1 || { tail -5 $root/gpurun_out/${tag}_hyper_once.txt; exit 1; }
f=$(ls $root/gpurun_out/prof_$tag/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY' | tee $root/gpurun_out/${tag}_hyper_kernels.txt
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    if 'sicn' in r['Name'] or 'anonymous' in r['Name']:
        tot += float(r['TotalDurationNs'])
        print(f"{r['Name'][:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
print(f"sum of kernel time per step (6 steps): {tot/6e6:.3f} ms")
PY
