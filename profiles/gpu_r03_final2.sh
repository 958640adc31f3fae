#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
bash profiles/gpu_r03_b.sh || exit $?
bash profiles/gpu_tuple_pmc.sh r03tup3 three || exit $?
bash profiles/gpu_tuple_pmc.sh r03tup2 two || exit $?
timeout -k 10 300 python3 profiles/exp_tuple.py > $OUT/r03_tuple_points.txt 2>&1; tail -6 $OUT/r03_tuple_points.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/r03_final_pytest.log 2>&1
rc=$?; tail -6 $OUT/r03_final_pytest.log; echo "pytest rc=$rc"
grep -l "Memory access fault" $OUT/r03_final_pytest.log && exit 9
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/r03_final_bench.json 2> $OUT/r03_final_bench.err || { tail -5 $OUT/r03_final_bench.err; exit 3; }
python3 -c "
import json;d=json.load(open('$OUT/r03_final_bench.json'))
print('value',d['value'],'roofline',{k:d['roofline'][k] for k in ('bound','achieved','peak','frac','frac_bounds','stale','measured_hbm_frac') if k in d['roofline']})
for s in d.get('secondary',[]): print(' sec',s['metric'][:50],s['value'],{k:s['roofline'].get(k) for k in ('frac','measured_hbm_frac','stale')})
"
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
