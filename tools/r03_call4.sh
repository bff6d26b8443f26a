#!/bin/bash
# round 3, GPU call 4: C4 with the symbol-sync tail on its own stream + host-built tap tables; new bench.py lines (c4 / c5 roofline)
set -u
export TMPDIR=/tmp
O=gpurun_out/r03d
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chan.py tests/test_gpu_sharding.py tests/test_gpu_modem_facade.py tests/test_gpu_dmo.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --config c4 --steps 20 --warmup 3 > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --config c5 --steps 20 --warmup 3 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o c4 -- python bench.py --config c4 --steps 5 --warmup 1 > $O/prof_c4.log 2>&1
for f in $(find $O/prof -name '*_results.db' | sort); do python tools/prof_summary.py $f "$(basename $f _results.db)"; done > $O/kernel_trace_summary.md 2>&1
find $O -name '*.csv' -size +2M -delete; find $O/prof -type f -size +4M -delete; find $O -name '*.db' -size +4M -delete
cat $O/bench_c4.json $O/bench_c5.json; tail -3 $O/bench_c4.err $O/bench_c5.err; cat $O/kernel_trace_summary.md
