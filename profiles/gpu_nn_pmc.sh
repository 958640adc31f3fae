#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-nnp}
AG=${2:-qr}
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --workload nn --nn-agents $AG --steps 40 --warmup 10 --no-cpu-baseline"
run_pmc () { local name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc_$name -- python3 $B > $OUT/${TAG}_pmc_$name.log 2>&1 || { echo "pmc $name failed"; exit 4; } }
run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
run_pmc waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_CYCLES
cd $ROOT && python3 profiles/pmc_summary.py $TAG --kernel ${3:-k_ptuple_episodes} | tail -22
