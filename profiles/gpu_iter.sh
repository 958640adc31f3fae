#!/bin/bash
# One GPU-box iteration of the round: parity tests first, then (only if no test process crashed)
# the bench lines and the counter passes.  Run through gpurun from the repo root:
#     gpurun --timeout 1100 -- 'bash profiles/gpu_iter.sh <tag> [tests] [pmc]'
TAG=${1:-it}
TESTS=${2:-"tests/test_gpu_parity.py tests/test_gpu_fuzz.py"}
PMC=${3:-yes}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 -m pytest $TESTS -x -q -m gpu > $OUT/${TAG}_pytest.log 2>&1
rc=$?
tail -15 $OUT/${TAG}_pytest.log
echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then echo "test process died (rc=$rc): no further GPU steps"; exit $rc; fi
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench20.json 2> $OUT/${TAG}_bench20.err || exit 3
cat $OUT/${TAG}_bench20.json | cut -c1-400
timeout -k 10 300 python3 bench.py --steps 100 --warmup 25 --no-cpu-baseline > $OUT/${TAG}_bench100.json 2> $OUT/${TAG}_bench100.err || exit 3
cut -c1-200 $OUT/${TAG}_bench100.json
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --epsilon 0.001 > $OUT/${TAG}_bench_late.json 2>> $OUT/${TAG}_bench20.err || exit 3
cut -c1-200 $OUT/${TAG}_bench_late.json
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype float64 > $OUT/${TAG}_bench_f64.json 2>> $OUT/${TAG}_bench20.err || exit 3
cut -c1-200 $OUT/${TAG}_bench_f64.json
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --noise-prob 0.05 > $OUT/${TAG}_bench_noise.json 2>> $OUT/${TAG}_bench20.err || exit 3
cut -c1-200 $OUT/${TAG}_bench_noise.json
if [ "$PMC" != "yes" ]; then exit 0; fi
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
run_pmc () {   # name counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc_$name -- python3 $B > $OUT/${TAG}_pmc_$name.log 2>&1 || { echo "pmc $name failed"; exit 4; }
}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $B > $OUT/${TAG}_stats.log 2>&1 || exit 4
run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
run_pmc waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_CYCLES
run_pmc active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH
run_pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL
run_pmc valu SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
echo "collected $TAG"
