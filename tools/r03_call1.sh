#!/bin/bash
# round 3, GPU call 1: LDS-DMA streaming microbenchmark, the whole GPU test suite (new: literal C4 form, TED contract macro), kernel
# traces of C4 / C5 as they stand (baseline for this round's kernel work)
set -u
export TMPDIR=/tmp
O=gpurun_out/r03a
rm -rf $O; mkdir -p $O
timeout 120 ./build/stream_lds 16 > $O/stream_lds.log 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for cfg in c4 c5; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o $cfg -- python bench.py --config $cfg --steps 5 --warmup 1 --no-extra > $O/prof_$cfg.log 2>&1
done
for f in $(find $O/prof -name '*_results.db' | sort); do python tools/prof_summary.py $f "$(basename $f _results.db)"; done > $O/kernel_trace_summary.md 2>&1
find $O -name '*.csv' -size +2M -delete; find $O/prof -type f -size +4M -delete; find $O -name '*.db' -size +4M -delete
tail -5 $O/pytest.log; cat $O/kernel_trace_summary.md | head -40; head -80 $O/stream_lds.log
