#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_nn.py -x -q > $OUT/nncheck_pytest.log 2>&1
rc=$?; tail -5 $OUT/nncheck_pytest.log; echo "pytest rc=$rc"
grep -l "Memory access fault" $OUT/nncheck_pytest.log && exit 9
[ $rc -ne 0 ] && exit $rc
bash profiles/gpu_nn_quick.sh nncheck "rr qr qa"
