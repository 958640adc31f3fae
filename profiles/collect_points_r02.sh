#!/bin/bash
# The other measurement points of SURVEY.md section 8(d), one JSON line each into gpurun_out/<tag>_points.txt:
#   gpurun --timeout 1150 -- 'bash profiles/collect_points_r02.sh r02'
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_points.txt
: > $OUT
run () { echo "## $*" >> $OUT; timeout -k 10 400 python3 $ROOT/bench.py "$@" --no-cpu-baseline 2>/dev/null | cut -c1-1500 >> $OUT || echo "FAILED" >> $OUT; }
run --steps 100 --warmup 25                                  # headline, 25 episodes per launch
run --steps 100 --warmup 25 --epsilon 0.001                  # late-training regime (greedy-dominated)
run --steps 100 --warmup 25 --noise-prob 0.05                # the class default noise_prob
run --steps 100 --warmup 25 --dtype float64                  # the reference's own table dtype, LDS-resident
run --steps 100 --warmup 25 --games 65536                    # BASELINE configs[1]
run --steps 100 --warmup 25 --games 65536 --dtype float64
run --steps 20 --warmup 5 --games 65536 --kernel generic     # the fallback kernel
run --steps 20 --warmup 5 --games 65536 --kernel generic --dtype float64
for w in 8 12 16 20; do
  echo "## waves/CU $w" >> $OUT
  THRL_WAVE_MAX_WAVES_PER_CU=$w timeout -k 10 300 python3 $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-400 >> $OUT
done
python3 - "$OUT" <<'PY'
import json, sys
name = None
for l in open(sys.argv[1]):
    if l.startswith("##"): name = l[2:].strip()
    elif l.startswith("{"):
        d = json.loads(l) if l.rstrip().endswith("}") else None
        if d: print("%-70s %.3e env-steps/s  %.2f ms/launch (%d episodes)" % (name, d["value"], d["roofline"]["avg_launch_ms"], d["config"]["episodes_per_launch"]))
        else: print(name, l[:120])
PY
