#!/bin/bash
# round 3, GPU call 5: k_decim_pm (phase-major matrix-pipe front end): whole GPU suite, C1 bench + kernel trace
set -u
export TMPDIR=/tmp
O=gpurun_out/r03e
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 15 $O/pytest.log
python bench.py --steps 20 --warmup 3 --no-extra --check > $O/bench_c1.json 2> $O/bench_c1.err
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o c1 -- python bench.py --config c1 --steps 5 --warmup 1 --no-extra > $O/prof_c1.log 2>&1
for f in $(find $O/prof -name '*_results.db' | sort); do python tools/prof_summary.py $f "$(basename $f _results.db)"; done > $O/kernel_trace_summary.md 2>&1
find $O -name '*.csv' -size +2M -delete; find $O/prof -type f -size +4M -delete; find $O -name '*.db' -size +4M -delete
cut -c1-1500 $O/bench_c1.json; tail -n 3 $O/bench_c1.err; cat $O/kernel_trace_summary.md
