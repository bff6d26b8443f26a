#!/bin/bash
# round 3, GPU call 37: W4 microbenchmark with conflict-free (padded) block strides in LDS
set -u
export TMPDIR=/tmp
O=gpurun_out/r03ak
rm -rf $O; mkdir -p $O
timeout 200 ./build/stream_lds 16 1 4 > $O/stream_w4_pad.log 2>&1
cat $O/stream_w4_pad.log
