#!/bin/bash
# round 3, second session: PMC / timing evidence on the final binary (headline at two launch sizes, neural pairings, tuple shapes)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash profiles/collect_r03.sh r03c25 25 || exit $?
bash profiles/collect_r03.sh r03c5 5 short || exit $?
bash profiles/collect_nn_r03.sh r03nn || exit $?
bash profiles/gpu_tuple_pmc.sh r03tup3 three || exit $?
bash profiles/gpu_tuple_pmc.sh r03tup2 two || exit $?
