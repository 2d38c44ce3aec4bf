#!/bin/bash
# Same-box A/B of one environment switch: bench.py --headline-only with and without it, interleaved.  usage: ab_env.sh SICN_NO_DEAL=1 [pairs]
sw="$1"; pairs="${2:-6}"
for round in $(seq 1 "$pairs"); do
  for leg in off on; do
    if [ "$leg" = on ]; then pre="$sw"; else pre="_UNUSED=0"; fi
    env "$pre" python bench.py --headline-only --steps 80 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$leg', '$sw', d['ms_per_step'], ' '.join('L%d=%.4f' % (l['layer'], l['ms']) for l in d['layers']), flush=True)"
  done
done
