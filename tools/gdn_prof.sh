#!/bin/bash
# k_gdn alone under rocprofv3 (tools/gdn_speed.py): per-kernel averages -> gpurun_out/<tag>_gdn_kernels.txt
# usage (on the GPU box, from the repo root): bash tools/gdn_prof.sh <tag>
tag=${1:-gdn}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/tools/gdn_speed.py > $root/gpurun_out/${tag}_gdn_speed.txt 2>&1
grep "^C=" $root/gpurun_out/${tag}_gdn_speed.txt
f=$(ls $root/gpurun_out/prof_$tag/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY' | tee $root/gpurun_out/${tag}_gdn_kernels.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'sicn' in r['Name']:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:9.1f} us")
PY
