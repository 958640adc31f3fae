#!/bin/bash
# A/B of two builds of libthrl_hip.so in ONE gpurun call (same box, back to back, three pairs), the way every
# "+x %" in DESIGN.md section 5.1 was measured:
#   cp th_rl_amd/libthrl_hip.so build/libthrl_base.so      # the build to compare against (build/ travels with gpurun)
#   ... edit, python -m th_rl_amd.build ...
#   gpurun --timeout 900 -- 'bash profiles/compare_builds.sh'
# THRL_LIB selects the library (th_rl_amd/_lib.py).  The wave-kernel parity tests run first on the new build.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wave" > $OUT/cmp_pytest.log 2>&1 || { tail -20 $OUT/cmp_pytest.log; exit 3; }
tail -1 $OUT/cmp_pytest.log
for lib in base new base new base new; do
  if [ $lib = base ]; then export THRL_LIB=$ROOT/build/libthrl_base.so; else unset THRL_LIB; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/cmp_${lib}.json 2>$OUT/cmp_${lib}.err || exit 4
  python3 -c "import json,sys; d=json.load(open('$OUT/cmp_${lib}.json')); print('$lib value %.4e median launch %.3f'%(d['value'], d['roofline']['median_launch_ms']))"
done
# trained tables: 10,000 episodes of the run first (GREEDY variants)
for lib in base new; do
  if [ $lib = base ]; then export THRL_LIB=$ROOT/build/libthrl_base.so; else unset THRL_LIB; fi
  timeout -k 10 300 python3 bench.py --games 262144 --steps 100 --warmup 25 --chunk 25 --no-cpu-baseline --pretrain 10000 > $OUT/cmp_${lib}.json 2>$OUT/cmp_${lib}.err || exit 4
  python3 -c "import json,sys; d=json.load(open('$OUT/cmp_${lib}.json')); print('$lib trained value %.4e'%(d['value']))"
done
