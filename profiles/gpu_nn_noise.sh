#!/bin/bash
# neural pairings with the environment class's default noise (0.05): the general episode kernel + the update kernel's
# one-state-per-transition path
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ag in rr qr; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nnnoise_${ag}_stats -- python3 $ROOT/bench.py --workload nn --nn-agents $ag --steps 40 --warmup 10 --no-cpu-baseline --noise-prob 0.05 > $OUT/nnnoise_${ag}.log 2>&1 || { tail -5 $OUT/nnnoise_${ag}.log; exit 4; }
  python3 - <<PY
import csv,glob,json
f=sorted(glob.glob('$OUT/nnnoise_${ag}_stats/*/*_kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:3]:
    print("  %-58s calls %s avg %.3f ms %s%%"%(r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e6, r['Percentage']))
l=[x for x in open('$OUT/nnnoise_${ag}.log') if x.startswith('{')][-1]
print("  $ag noise 0.05: value %.4g (under rocprof)"%json.loads(l)['value'])
PY
done
