#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03ai
rm -rf $O; mkdir -p $O
for i in 1 2; do
  timeout 300 python -m pytest tests/test_gpu_chan.py tests/test_gpu_sharding.py -x -q 2>&1 | tail -1 >> $O/log.txt
  QRL_LIB_PATH=$PWD/build/libqrl_ctold.so timeout 300 python -m pytest tests/test_gpu_chan.py tests/test_gpu_sharding.py -x -q 2>&1 | tail -1 >> $O/log.txt
done
timeout 300 python -m pytest tests/test_gpu_chan.py tests/test_gpu_sharding.py -x -q 2>&1 | grep -E "^E   |assert" | head -20 >> $O/log.txt
cat $O/log.txt
