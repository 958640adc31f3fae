#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
bash profiles/collect_nn_r03.sh r03nn || exit $?
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/r03_final_pytest.log 2>&1
rc=$?; tail -6 $OUT/r03_final_pytest.log; echo "pytest rc=$rc"
grep -l "Memory access fault" $OUT/r03_final_pytest.log && exit 9
exit $rc
