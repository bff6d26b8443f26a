#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03g
rm -rf $O; mkdir -p $O
QRL_LIB_PATH=$PWD/build/libqrl_pmprof.so python tools/pm_prof.py > $O/pm_prof.log 2>&1
cat $O/pm_prof.log
