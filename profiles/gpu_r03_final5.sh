#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
bash profiles/collect_nn_r03.sh r03nn || exit $?
bash profiles/gpu_tuple_pmc.sh r03tup3 three || exit $?
bash profiles/gpu_tuple_pmc.sh r03tup2 two || exit $?
cd $ROOT
timeout -k 10 300 python3 profiles/exp_tuple.py > $OUT/r03_tuple_points.txt 2>&1; tail -5 $OUT/r03_tuple_points.txt
