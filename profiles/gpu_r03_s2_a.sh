#!/bin/bash
# round 3, second session: whole GPU suite, smoke, default bench line, tuple points, tuple ablation
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/r03_final_pytest.log 2>&1
rc=$?; tail -6 $OUT/r03_final_pytest.log; echo "pytest rc=$rc"
grep -l "Memory access fault" $OUT/r03_final_pytest.log && exit 9
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 5
timeout -k 10 600 python3 bench.py > $OUT/r03_bench.json 2> $OUT/r03_bench.err || { tail -5 $OUT/r03_bench.err; exit 6; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench.json").read().strip().splitlines()[-1])
print("headline %.4g frac %.3f stale %s" % (d["value"], d["roofline"]["frac"], d["roofline"].get("stale")))
for s in d.get("secondary", []):
    print("  %s: %.4g  stale %s" % (s["metric"], s["value"], s["roofline"].get("stale")))
PY
timeout -k 10 300 python3 profiles/exp_tuple.py > $OUT/r03_tuple_points.txt 2>&1; tail -9 $OUT/r03_tuple_points.txt
timeout -k 10 300 python3 profiles/ablate_tuple.py three > $OUT/r03_tuple_ablate_three.txt 2>&1; cat $OUT/r03_tuple_ablate_three.txt
timeout -k 10 300 python3 profiles/ablate_tuple.py two > $OUT/r03_tuple_ablate_two.txt 2>&1; cat $OUT/r03_tuple_ablate_two.txt
