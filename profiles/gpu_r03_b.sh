#!/bin/bash
# round 3, GPU call B: the issue microbenchmark, alone and under the SQ counters that the kernel's roofline uses
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
set -o pipefail
timeout -k 10 400 ./build/ubench_issue 4000 > $OUT/r03_ubench_issue.json 2> $OUT/r03_ubench_issue.err || { echo "ubench failed"; tail -3 $OUT/r03_ubench_issue.err; exit 5; }
if grep -q "Memory access fault" $OUT/r03_ubench_issue.err; then echo "GPU fault"; exit 6; fi
wc -c $OUT/r03_ubench_issue.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS \
    --kernel-trace --output-format csv -d $OUT/r03_ubench_pmc -- $ROOT/build/ubench_issue 2000 > $OUT/r03_ubench_pmc.json 2> $OUT/r03_ubench_pmc.err || { echo "ubench pmc failed"; tail -5 $OUT/r03_ubench_pmc.err; exit 7; }
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC \
    --kernel-trace --output-format csv -d $OUT/r03_ubench_pmc2 -- $ROOT/build/ubench_issue 2000 > $OUT/r03_ubench_pmc2.json 2> $OUT/r03_ubench_pmc2.err || { echo "ubench pmc2 failed"; tail -5 $OUT/r03_ubench_pmc2.err; exit 7; }
find $OUT/r03_ubench_pmc $OUT/r03_ubench_pmc2 -name "*.csv" | head; du -sh $OUT/r03_ubench_pmc $OUT/r03_ubench_pmc2
