#!/bin/bash
# A/B of two bench configurations inside ONE gpurun call (same box, interleaved):  tools/ab_bench.sh "ENV_A=.." "ENV_B=.." [reps]
A="$1"; B="$2"; R=${3:-3}
for r in $(seq $R); do
  for cfg in "$A" "$B"; do
    env $cfg timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-sequence --no-strong 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg', 'it/s %.1f sustained %.1f conv frac %.3f (%.1f us) rays/s %.2fM corr %.1f us (%.3f)' % (d['value'], d['sustained_value'], d['roofline']['frac'], 1e3*d['roofline']['ms_per_launch'], d['rays_per_sec']/1e6, 1e3*d['roofline_corr']['ms_per_launch'], d['roofline_corr']['frac']))"
  done
done
