#!/bin/bash
# PMC rows of every kernel of the hyperprior step (8 x 4K, tools/hyper_once.py): what bounds k_gdn and k_l0g.  Three passes, each
# --kernel-trace + --pmc only (SQ counters; FETCH_SIZE; WRITE_SIZE).  usage (GPU box, repo root): bash tools/hyper_pmc.sh <tag>
# -> gpurun_out/<tag>_hyper_pmc.txt (tools/hyper_pmc_summary.py)
tag=${1:-hpmc}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$out/sq" -o pmc --output-format csv -- python3 $root/tools/hyper_once.py > /dev/null 2> "$out/sq.err" &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/fetch" -o pmc --output-format csv -- python3 $root/tools/hyper_once.py > /dev/null 2> "$out/fetch.err" &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/write" -o pmc --output-format csv -- python3 $root/tools/hyper_once.py > /dev/null 2> "$out/write.err" &&
python3 $root/tools/hyper_pmc_summary.py "$out" | tee $root/gpurun_out/${tag}_hyper_pmc.txt
