#!/bin/bash
# noise_prob 0.05 (the environment's default) at fewer resident waves per CU (THRL_WAVE_MAX_WAVES_PER_CU): what would 20 waves be worth?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_noise_occupancy.txt
cd $ROOT
: > $OUT
for w in 5 10 15; do
  THRL_WAVE_MAX_WAVES_PER_CU=$w timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --noise-prob 0.05 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('noise_prob 0.05, at most $w waves per CU: %.4g env-steps/s  launch %.2f ms' % (d['value'], d['roofline'].get('avg_launch_ms', 0)))" >> $OUT
done
cat $OUT
