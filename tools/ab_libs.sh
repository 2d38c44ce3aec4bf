#!/bin/bash
# Same-box comparison of library builds: bench each lib twice, interleaved. usage: ab_libs.sh lib1.so lib2.so ...
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    SICN_LIB=$PWD/$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['ms_per_step'], ' '.join('L%d=%.3f' % (l['layer'], l['ms']) for l in d['layers']))"
  done
done
