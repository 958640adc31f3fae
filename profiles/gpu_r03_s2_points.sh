#!/bin/bash
# other points of SURVEY 8(d) on the final binary: BASELINE configs[1] size, late-training regime, float64 tables, default env noise
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_other_points.txt
cd $ROOT
: > $OUT
run () { local label=$1; shift
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %.4g env-steps/s  %.3f ms/step  kernel %s  launch %.2f ms' % ('$label', d['value'], d['ms_per_step'], d['config'].get('kernel'), d['roofline'].get('avg_launch_ms', 0)))" >> $OUT || echo "$label FAILED" >> $OUT
}
run "configs[1]: 65,536 games" --games 65536
run "2^20 games (configs[2], the headline)"
run "late training (epsilon 0.001)" --epsilon 0.001
run "float64 tables" --dtype float64
run "noise_prob 0.05 (environment default)" --noise-prob 0.05
run "noise_prob 0.05, float64" --noise-prob 0.05 --dtype float64
run "max_steps 50 (training cycle of 2 episodes)" --max-steps 50 --warmup 6
cat $OUT
