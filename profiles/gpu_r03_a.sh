#!/bin/bash
# round 3, GPU call A: new parity tests, the "does the new test catch the round-2 bug" check, full suite,
# issue microbenchmark, bench lines (incl. the --gpus 2 launcher branch)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_regimes.py -x -q -m gpu > $OUT/r03a_regimes.log 2>&1
rc=$?; tail -5 $OUT/r03a_regimes.log; echo "regimes rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
THRL_LIB=$ROOT/build/libthrl_bug_same4.so timeout -k 10 300 python3 -m pytest tests/test_gpu_regimes.py -q -m gpu -k "noisy_step_inside" > $OUT/r03a_bugteeth.log 2>&1
echo "bug-teeth rc=$? (expected 1: the round-2 shortcut must FAIL the new test)"; tail -4 $OUT/r03a_bugteeth.log
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/r03a_pytest.log 2>&1
rc=$?; tail -5 $OUT/r03a_pytest.log; echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 ./build/ubench_issue 4000 > $OUT/r03_ubench_issue.json 2> $OUT/r03_ubench_issue.err || { echo ubench failed; tail -3 $OUT/r03_ubench_issue.err; }
wc -c $OUT/r03_ubench_issue.json
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/r03a_bench20.json 2> $OUT/r03a_bench20.err || exit 3
cut -c1-300 $OUT/r03a_bench20.json
timeout -k 10 120 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/r03a_bench_g2_refuse.json 2> $OUT/r03a_bench_g2_refuse.err
echo "gpus 2 plain rc=$? (expected 2)"; cat $OUT/r03a_bench_g2_refuse.err | tail -2
timeout -k 10 300 python3 bench.py --gpus 2 --allow-oversubscribe --steps 20 --warmup 5 --games 262144 > $OUT/r03a_bench_g2.json 2> $OUT/r03a_bench_g2.err
echo "gpus 2 oversubscribed rc=$?"; cut -c1-300 $OUT/r03a_bench_g2.json; tail -3 $OUT/r03a_bench_g2.err
