#!/bin/bash
# Round-2 evidence for BASELINE configs[3] (neural policies x 65,536 games):
#   gpurun --timeout 1150 -- 'bash profiles/collect_nn_r02.sh r02nn'
TAG=${1:-r02nn}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
for p in rr qr qa qc; do
  timeout -k 10 300 python3 $ROOT/bench.py --workload nn --nn-agents $p --steps 20 --warmup 10 > $OUT/${TAG}_${p}_bench.json 2> $OUT/${TAG}_${p}_bench.err || exit 3
  cut -c1-160 $OUT/${TAG}_${p}_bench.json
done
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --workload nn --nn-agents rr --steps 20 --warmup 10 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $B > $OUT/${TAG}_stats.log 2>&1 || exit 4
run_pmc () {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc_$name -- python3 $B > $OUT/${TAG}_pmc_$name.log 2>&1 || { echo "pmc $name failed"; exit 4; }
}
run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
run_pmc waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_CYCLES
run_pmc active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
B2="$ROOT/bench.py --workload nn --nn-agents qa --steps 20 --warmup 10 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_qa_stats -- python3 $B2 > $OUT/${TAG}_qa_stats.log 2>&1 || exit 4
B3="$ROOT/bench.py --workload nn --nn-agents qr --steps 20 --warmup 10 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_qr_stats -- python3 $B3 > $OUT/${TAG}_qr_stats.log 2>&1 || exit 4
B4="$ROOT/bench.py --workload nn --nn-agents qc --steps 20 --warmup 10 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_qc_stats -- python3 $B4 > $OUT/${TAG}_qc_stats.log 2>&1 || exit 4
echo "collected $TAG"
