#!/bin/bash
# SQ counters of k_wave_episodes on TRAINED tables (bench.py --pretrain): what changes between the exploring
# regime the headline is quoted on and the greedy regime a 20,000-episode run spends most of its time in.
#   gpurun --timeout 1100 -- 'bash profiles/collect_late_r02.sh <tag> <pretrain-episodes>'
# (summaries: python profiles/pmc_summary.py <tag> --last 4)
TAG=${1:-late}
PRE=${2:-10000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --games 262144 --steps 100 --warmup 25 --chunk 25 --no-cpu-baseline --pretrain $PRE"
run_pmc () {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc_$name -- python3 $B > $OUT/${TAG}_pmc_$name.log 2>&1 || { echo "pmc $name failed"; exit 4; }
}
timeout -k 10 300 python3 $B > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 3
run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
run_pmc waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_CYCLES
run_pmc active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH
run_pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL
echo "collected $TAG pretrain $PRE"
