#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03l
rm -rf $O; mkdir -p $O
timeout 200 ./build/stream_lds 16 1 3 > $O/stream_w4.log 2>&1
cat $O/stream_w4.log
