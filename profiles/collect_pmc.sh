#!/bin/bash
# Collects the per-round profile evidence on the GPU box (run through gpurun from the repo root):
#   profiles/collect_pmc.sh r04
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py --headline-only --steps 40` — the timed loop + the per-layer loop and
#    nothing else on the GPU (round 2 profiled the secondary legs too, and the per-layer averages mixed four contexts)
# 2. three separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ counters in a third), each with
#    --kernel-trace only (never combined with sys/hip/hsa tracing: gpurun refuses that)
# Raw outputs go to gpurun_out/<tag>_prof/ (scratch); profiles/summarize_pmc.py turns them into the tracked
# profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_summary.json (with the kernel-source fingerprint bench.py checks).
set -e -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_prof
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --headline-only --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- python3 bench.py --headline-only --sustained --steps 40 --warmup 5 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d "$OUT/pmc_fetch" -o pmc --output-format csv -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d "$OUT/pmc_write" -o pmc --output-format csv -- $BENCH > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o pmc --output-format csv -- $BENCH > /dev/null 2> "$OUT/pmc_sq.err"
python3 profiles/summarize_pmc.py "$TAG" "$OUT"
