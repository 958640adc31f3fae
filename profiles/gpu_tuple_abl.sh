#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-tabl}
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_tuple.py tests/test_gpu_nn.py -x -q -m gpu > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -8 $OUT/${TAG}_pytest.log; echo "pytest rc=$rc"
grep -l "Memory access fault" $OUT/${TAG}_pytest.log && exit 9
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 profiles/ablate_tuple.py three > $OUT/${TAG}_three.txt 2>&1; cat $OUT/${TAG}_three.txt
timeout -k 10 300 python3 profiles/ablate_tuple.py two > $OUT/${TAG}_two.txt 2>&1; cat $OUT/${TAG}_two.txt
