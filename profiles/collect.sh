#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#     gpurun --timeout 1100 -- 'bash profiles/collect.sh r01'
# then, back in the container:  python profiles/summarize.py r01
# Kernel timing and each PMC counter are separate passes (the pool refuses --pmc mixed with traces).
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py --steps 100 --warmup 25"      # every launch = 25 episodes, so per-dispatch means are clean
python3 $BENCH > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $BENCH --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $BENCH --no-cpu-baseline > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $BENCH --no-cpu-baseline > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/${TAG}_pmc_insts -- python3 $BENCH --no-cpu-baseline > $OUT/${TAG}_pmc_insts.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_pmc_waves -- python3 $BENCH --no-cpu-baseline > $OUT/${TAG}_pmc_waves.log 2>&1
python3 $ROOT/bench.py --workload nn --steps 20 --warmup 10 > $OUT/${TAG}_nn_bench.json 2>> $OUT/${TAG}_bench.err
python3 $ROOT/bench.py --workload nn --nn-agents qr --steps 20 --warmup 10 > $OUT/${TAG}_nnqr_bench.json 2>> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_nn_stats -- python3 $ROOT/bench.py --workload nn --steps 20 --warmup 10 > $OUT/${TAG}_nn_stats.log 2>&1

python3 $ROOT/profiles/run_example_config.py > $OUT/${TAG}_example_config.log 2>&1
echo collected $TAG
