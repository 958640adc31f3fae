#!/bin/bash
# Round-3 evidence for BASELINE configs[3] (neural policies x 65,536 games): bench lines of the four pairings, and for
# 2 x Reinforce (rr), QTable vs Reinforce (qr) and QTable vs ActorCritic (qa) the kernel timing + the PMC passes nn_traffic.json is built from.
#   gpurun --timeout 1150 -- 'bash profiles/collect_nn_r03.sh r03nn'
TAG=${1:-r03nn}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python3 -c "from th_rl_amd import _lib; import json; print(json.dumps(_lib.build_info()))" > $OUT/${TAG}_library.json || exit 2
for p in rr qr qa qc; do
  timeout -k 10 300 python3 $ROOT/bench.py --workload nn --nn-agents $p --steps 40 --warmup 10 > $OUT/${TAG}_${p}_bench.json 2> $OUT/${TAG}_${p}_bench.err || exit 3
  cut -c1-160 $OUT/${TAG}_${p}_bench.json
done
cd /tmp && export TMPDIR=/tmp
for p in rr qr qa; do
  B="$ROOT/bench.py --workload nn --nn-agents $p --steps 40 --warmup 10 --no-cpu-baseline"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${p}_stats -- python3 $B > $OUT/${TAG}_${p}_stats.log 2>&1 || exit 4
  run_pmc () {
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_${p}_pmc_$name -- python3 $B > $OUT/${TAG}_${p}_pmc_$name.log 2>&1 || { echo "pmc $p $name failed"; tail -3 $OUT/${TAG}_${p}_pmc_$name.log; exit 4; }
    grep -l "Memory access fault" $OUT/${TAG}_${p}_pmc_$name.log && exit 9
  }
  run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
  run_pmc waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_CYCLES
  run_pmc fetch FETCH_SIZE
  run_pmc write WRITE_SIZE
done
echo "collected $TAG"
