#!/bin/bash
# Round-3 evidence for the dominant kernel (k_wave_episodes<float,2,1>) at ONE launch size:
#   gpurun --timeout 1150 -- 'bash profiles/collect_r03.sh r03c25 25'      (and r03c5 5 for the traffic model)
# Kernel timing and every PMC group are SEPARATE rocprofv3 passes (the pool refuses --pmc with traces); FETCH_SIZE and
# WRITE_SIZE are separate passes too (TCC slots).  profiles/summarize_r03.py turns the raw output into the committed
# summaries and profiles/traffic.json (with the hash of the library the counters were collected on).
TAG=${1:-r03c25}
CH=${2:-25}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python3 -c "from th_rl_amd import _lib; import json; print(json.dumps(_lib.build_info()))" > $OUT/${TAG}_library.json || exit 2
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --steps $((CH*4)) --warmup $CH --chunk $CH --no-cpu-baseline --no-secondary"
run_pmc () {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc_$name -- python3 $B > $OUT/${TAG}_pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $OUT/${TAG}_pmc_$name.log; exit 4; }
  grep -l "Memory access fault" $OUT/${TAG}_pmc_$name.log && exit 9
}
timeout -k 10 300 python3 $B > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $B > $OUT/${TAG}_stats.log 2>&1 || exit 4
run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
run_pmc waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_CYCLES
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
if [ "$3" != "short" ]; then
run_pmc clock GRBM_GUI_ACTIVE
run_pmc active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH
run_pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL
run_pmc valu SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT
fi
echo "collected $TAG chunk $CH"
