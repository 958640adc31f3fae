#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-full}
cd $ROOT
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -8 $OUT/${TAG}_pytest.log; echo "pytest rc=$rc"
grep -l "Memory access fault" $OUT/${TAG}_pytest.log && exit 9
exit $rc
