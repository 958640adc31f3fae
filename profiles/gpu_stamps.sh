#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
THRL_LIB=$ROOT/build/libthrl_stamp.so timeout -k 10 300 python3 profiles/exp_train_stamps.py rr 2>&1 | tail -10

