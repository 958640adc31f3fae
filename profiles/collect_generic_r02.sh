#!/bin/bash
# rocprofv3 passes for the fallback kernel k_generic_episodes (65,536 games, float32):
#   gpurun --timeout 900 -- 'bash profiles/collect_generic_r02.sh r02gen'
TAG=${1:-r02gen}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --kernel generic --games 65536 --steps 20 --warmup 5 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $B > $OUT/${TAG}_stats.log 2>&1 || exit 4
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -- python3 $B > $OUT/${TAG}_pmc_$c.log 2>&1 || exit 4
done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 $B > $OUT/${TAG}_pmc_sq.log 2>&1 || exit 4
echo collected $TAG
