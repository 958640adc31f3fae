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
run --steps 100 --warmup 25 --games 262144 --pretrain 4000   # trained tables (GREEDY variant once epsilon <= 0.17)
run --steps 100 --warmup 25 --games 262144 --pretrain 10000
run --steps 100 --warmup 25 --games 262144 --pretrain 20000
run --steps 96 --warmup 24 --chunk 24 --max-steps 50         # training cycles: 2 episodes per train_net
run --steps 90 --warmup 30 --chunk 30 --max-steps 10         # 10 episodes per train_net
run --steps 100 --warmup 25 --capacity 64                    # 36 of 100 transitions dropped from the deque
for w in 8 12 16 20; do
  echo "## waves/CU $w" >> $OUT
  THRL_WAVE_MAX_WAVES_PER_CU=$w timeout -k 10 300 python3 $ROOT/bench.py --steps 40 --warmup 10 --chunk 10 --regions 1 --no-cpu-baseline 2>/dev/null | cut -c1-400 >> $OUT
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
# occupancy fit 1/throughput = a/W + b (copy gpurun_out/<tag>_wave_scaling.json to profiles/r02_wave_scaling.json)
python3 - "$OUT" > $ROOT/gpurun_out/${TAG}_wave_scaling.json <<'PY2'
import json, re, sys
import numpy as np
W, V, name = [], [], None
for l in open(sys.argv[1]):
    if l.startswith("## waves/CU"): name = int(l.split()[-1])
    elif l.startswith("{") and name:
        m = re.search(r'"value": ([0-9.e+]+)', l)
        if m: W.append(name); V.append(float(m.group(1)))
        name = None
A = np.stack([1.0 / np.array(W, float), np.ones(len(W))], 1)
(a, b), *_ = np.linalg.lstsq(A, 1.0 / np.array(V), rcond=None)
share = (a / W[-1]) / (a / W[-1] + b)
print(json.dumps({"what": "env-steps/s vs resident waves per CU (THRL_WAVE_MAX_WAVES_PER_CU), 2^20 games, 10 episodes per launch",
                  "waves_per_cu": W, "env_steps_per_s": V,
                  "fit": "1/throughput = a/W + b with a = %.3g, b = %.3g: at W = %d the per-wave latency term a/W is %.0f %% of the time per env-step" % (a, b, W[-1], 100 * share),
                  "latency_share_at_20_waves": round(float(share), 3)}, indent=1))
PY2
