#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-tup}
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_tuple.py -x -q -m gpu > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -25 $OUT/${TAG}_pytest.log; echo "pytest rc=$rc"
grep -l "Memory access fault" $OUT/${TAG}_pytest.log && exit 9
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 profiles/exp_tuple.py > $OUT/${TAG}_exp.log 2>&1; tail -12 $OUT/${TAG}_exp.log
