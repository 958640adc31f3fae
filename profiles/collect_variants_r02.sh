#!/bin/bash
# rocprofv3 --kernel-trace --stats passes for the other variants of the wave kernel (2^20 games, 25 episodes per launch):
#   gpurun --timeout 900 -- 'bash profiles/collect_variants_r02.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run () { local tag=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02v_$tag -- python3 $ROOT/bench.py --steps 100 --warmup 25 --no-cpu-baseline "$@" > $OUT/r02v_$tag.log 2>&1 || exit 4; }
run f64 --dtype float64
run noise --noise-prob 0.05
run late --epsilon 0.001
run cycle2 --max-steps 50 --steps 96 --warmup 24 --chunk 24
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/r02v_f64_pmc_$c -- python3 $ROOT/bench.py --steps 100 --warmup 25 --no-cpu-baseline --dtype float64 > $OUT/r02v_f64_pmc_$c.log 2>&1 || exit 4
done
echo collected variants
