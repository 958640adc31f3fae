#!/bin/bash
# round 3, GPU call C: ubench with the extra classes; the neural suites on the game-major replay rings; nn bench lines + kernel stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
set -o pipefail
fault () { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT"; exit 9; }; return 0; }
timeout -k 10 400 ./build/ubench_issue 4000 > $OUT/r03_ubench_issue.json 2> $OUT/r03_ubench_issue.err || { echo "ubench failed"; tail -3 $OUT/r03_ubench_issue.err; exit 5; }
fault $OUT/r03_ubench_issue.err
timeout -k 10 900 python3 -m pytest tests/test_gpu_nn.py tests/test_gpu_api.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/r03c_pytest.log 2>&1
rc=$?; tail -5 $OUT/r03c_pytest.log; echo "pytest rc=$rc"
fault $OUT/r03c_pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
for ag in rr qr qa qc; do
  timeout -k 10 300 python3 bench.py --workload nn --nn-agents $ag --steps 40 --warmup 10 --no-cpu-baseline > $OUT/r03c_nn${ag}_bench.json 2> $OUT/r03c_nn${ag}_bench.err || { echo "nn bench $ag failed"; tail -3 $OUT/r03c_nn${ag}_bench.err; exit 3; }
  cut -c1-220 $OUT/r03c_nn${ag}_bench.json
done
cd /tmp && export TMPDIR=/tmp
for ag in rr qr; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03c_nn${ag}_stats -- python3 $ROOT/bench.py --workload nn --nn-agents $ag --steps 40 --warmup 10 --no-cpu-baseline > $OUT/r03c_nn${ag}_stats.log 2>&1 || { echo "stats $ag failed"; exit 4; }
done
find $OUT/r03c_nnrr_stats -name "*kernel_stats.csv" | head -2
