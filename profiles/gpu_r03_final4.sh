#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/r03_final_bench.json 2> $OUT/r03_final_bench.err || { tail -5 $OUT/r03_final_bench.err; exit 3; }
python3 -c "
import json;d=json.load(open('$OUT/r03_final_bench.json'))
print('value',d['value'],'roofline',{k:d['roofline'][k] for k in ('bound','achieved','peak','frac','frac_bounds','stale','measured_hbm_frac','salu_frac','lds_frac') if k in d['roofline']})
for s in d.get('secondary',[]): print(' sec',s['metric'][:50],s['value'],{k:s['roofline'].get(k) for k in ('frac','measured_hbm_frac','stale')})
"
timeout -k 10 300 python3 bench.py --steps 100 --warmup 25 --no-cpu-baseline --no-secondary > $OUT/r03_final_bench100.json 2>> $OUT/r03_final_bench.err || exit 3
python3 -c "import json;d=json.load(open('$OUT/r03_final_bench100.json'));print('steps100', d['value'], d['roofline']['frac'])"
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 200 python3 bench.py --gpus 2 --steps 20 --warmup 5 > /dev/null 2> $OUT/r03_final_g2.err; echo "gpus2 plain rc=$?"; tail -1 $OUT/r03_final_g2.err
