#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03h
rm -rf $O; mkdir -p $O
timeout 200 ./build/stream_lds 16 1 2 > $O/stream_w3.log 2>&1
cat $O/stream_w3.log
