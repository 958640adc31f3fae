#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/r03e_bench_default.json 2> $OUT/r03e_bench_default.err || { echo "default bench failed"; tail -20 $OUT/r03e_bench_default.err; exit 3; }
python3 -c "
import json;d=json.load(open('$OUT/r03e_bench_default.json'))
print('value',d['value'],'roofline',{k:d['roofline'][k] for k in ('bound','achieved','peak','frac','stale') if k in d['roofline']})
for s in d.get('secondary',[]): print(' sec',s['metric'][:50],s['value'],s['roofline'].get('frac'))
"
bash profiles/collect_r03.sh r03c25 25 || exit $?
bash profiles/collect_r03.sh r03c5 5 short || exit $?
