#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-tupb}
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_tuple.py tests/test_gpu_nn.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -5 $OUT/${TAG}_pytest.log; echo "pytest rc=$rc"
grep -l "Memory access fault" $OUT/${TAG}_pytest.log && exit 9
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 profiles/exp_tuple.py > $OUT/${TAG}_exp.log 2>&1; tail -5 $OUT/${TAG}_exp.log
for ag in rr qr; do timeout -k 10 200 python3 bench.py --workload nn --nn-agents $ag --steps 40 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_$ag.json 2>/dev/null; python3 -c "import json;print('$ag', json.load(open('$OUT/${TAG}_$ag.json'))['value'])"; done
