#!/bin/bash
# round 3, GPU call 26: kernel timeline of C4 and C1 steps (which kernels overlap, where the gaps are)
set -u
export TMPDIR=/tmp
O=gpurun_out/r03z
rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace -d $O/p4 -o c4 -- python bench.py --config c4 --steps 4 --warmup 1 --no-extra > $O/p4.log 2>&1
f=$(find $O/p4 -name '*_results.db' | head -1); python tools/prof_timeline.py $f 40 > $O/timeline_c4.txt 2>&1
timeout 200 rocprofv3 --kernel-trace -d $O/p1 -o c1 -- python bench.py --config c1 --steps 4 --warmup 1 --no-extra > $O/p1.log 2>&1
f=$(find $O/p1 -name '*_results.db' | head -1); python tools/prof_timeline.py $f 40 > $O/timeline_c1.txt 2>&1
find $O -name '*.db' -delete; find $O -name '*.csv' -size +1M -delete
cat $O/timeline_c4.txt; cat $O/timeline_c1.txt
