#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_tuple.py -x -q > $OUT/tuple_noise_pytest.log 2>&1
rc=$?; tail -15 $OUT/tuple_noise_pytest.log; echo "pytest rc=$rc"
grep -l "Memory access fault" $OUT/tuple_noise_pytest.log && exit 9
exit $rc
