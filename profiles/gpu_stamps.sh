#!/bin/bash
# phase timestamps of the network update kernel from the DIAGNOSTIC build build/libthrl_stamp.so (see exp_train_stamps.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for w in ${1:-rr qr}; do
  THRL_LIB=$ROOT/build/libthrl_stamp.so timeout -k 10 300 python3 profiles/exp_train_stamps.py $w 2>&1 | tail -15
done
