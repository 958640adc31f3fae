#!/bin/bash
# round 3, GPU call D: neural suites + nn bench lines with kernel stats (after a kernel change); optional ubench refresh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-r03d}
mkdir -p $OUT
cd $ROOT
set -o pipefail
fault () { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT"; exit 9; }; return 0; }
timeout -k 10 900 python3 -m pytest tests/test_gpu_nn.py tests/test_gpu_api.py -x -q -m gpu > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -5 $OUT/${TAG}_pytest.log; echo "pytest rc=$rc"
fault $OUT/${TAG}_pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
if [ "$2" = "ubench" ]; then
  timeout -k 10 400 ./build/ubench_issue 4000 > $OUT/r03_ubench_issue.json 2> $OUT/r03_ubench_issue.err || { echo "ubench failed"; exit 5; }
  fault $OUT/r03_ubench_issue.err
fi
cd /tmp && export TMPDIR=/tmp
for ag in rr qr; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_nn${ag}_stats -- python3 $ROOT/bench.py --workload nn --nn-agents $ag --steps 40 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_nn${ag}_stats.log 2>&1 || { echo "stats $ag failed"; tail -5 $OUT/${TAG}_nn${ag}_stats.log; exit 4; }
  python3 - <<PY
import csv,glob,json
f=sorted(glob.glob('$OUT/${TAG}_nn${ag}_stats/*/*_kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:3]:
    print("  %-58s calls %s avg %.3f ms %s%%"%(r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e6, r['Percentage']))
l=[x for x in open('$OUT/${TAG}_nn${ag}_stats.log') if x.startswith('{')][-1]
print("  $ag value %.4g (under rocprof)"%json.loads(l)['value'])
PY
done
