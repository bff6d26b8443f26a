#!/bin/bash
# round 3, GPU call 2: k_decim_pl2 (LDS-DMA front end) and k_pfb_chan64 -- parity, then A/B timing against the kernels they replace
set -u
export TMPDIR=/tmp
O=gpurun_out/r03b
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --steps 20 --warmup 3 --no-extra > $O/bench_c1.json 2> $O/bench_c1.err
python bench.py --steps 20 --warmup 3 --no-extra --legacy-frontend > $O/bench_c1_legacy.json 2> $O/bench_c1_legacy.err
python bench.py --config c4 --steps 20 --warmup 3 --no-extra > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --config c4 --steps 20 --warmup 3 --no-extra --legacy-pfb > $O/bench_c4_legacy.json 2> $O/bench_c4_legacy.err
for cfg in c1 c4; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o $cfg -- python bench.py --config $cfg --steps 5 --warmup 1 --no-extra > $O/prof_$cfg.log 2>&1
done
for f in $(find $O/prof -name '*_results.db' | sort); do python tools/prof_summary.py $f "$(basename $f _results.db)"; done > $O/kernel_trace_summary.md 2>&1
find $O -name '*.csv' -size +2M -delete; find $O/prof -type f -size +4M -delete; find $O -name '*.db' -size +4M -delete
cat $O/bench_c1.json $O/bench_c1_legacy.json $O/bench_c4.json $O/bench_c4_legacy.json | cut -c1-400; cat $O/kernel_trace_summary.md
