#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-nnq}
cd /tmp && export TMPDIR=/tmp
for ag in ${2:-qr}; do
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
