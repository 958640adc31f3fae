#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash profiles/collect_r03.sh r03c25 25 || exit $?
bash profiles/collect_r03.sh r03c5 5 short || exit $?
