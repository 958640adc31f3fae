#!/bin/bash
# quick check: the wave-kernel parity tests + the driver-shape bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-q}
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_regimes.py -x -q -m gpu > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -3 $OUT/${TAG}_pytest.log; echo "pytest rc=$rc"
grep -l "Memory access fault" $OUT/${TAG}_pytest.log && exit 9
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench20.json 2> $OUT/${TAG}_bench20.err || exit 3
python3 -c "import json;d=json.load(open('$OUT/${TAG}_bench20.json'));print('driver shape', d['value'], d['region_ms'])"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --epsilon 0.001 > $OUT/${TAG}_bench_late.json 2>> $OUT/${TAG}_bench20.err || exit 3
python3 -c "import json;d=json.load(open('$OUT/${TAG}_bench_late.json'));print('late', d['value'])"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype float64 > $OUT/${TAG}_bench_f64.json 2>> $OUT/${TAG}_bench20.err || exit 3
python3 -c "import json;d=json.load(open('$OUT/${TAG}_bench_f64.json'));print('f64', d['value'])"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --noise-prob 0.05 > $OUT/${TAG}_bench_noise.json 2>> $OUT/${TAG}_bench20.err || exit 3
python3 -c "import json;d=json.load(open('$OUT/${TAG}_bench_noise.json'));print('noise', d['value'])"
