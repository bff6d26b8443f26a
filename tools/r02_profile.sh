#!/bin/bash
# Round-2 profile pass (run on the GPU box through gpurun): kernel-trace stats of the bench commands, HBM-traffic PMC passes of
# the front-end kernels (separate rocprofv3 runs, kernel-trace only, as MI355X_MICROARCH.md prescribes), bench lines, B sweep.
# Everything under gpurun_out/r02/; tools/r02_collect.py turns it into profiles/r02_*.
set -u
export TMPDIR=/tmp
O=gpurun_out/r02
rm -rf $O; mkdir -p $O
for cfg in c1 c2 c3 c5; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o $cfg -- python bench.py --config $cfg --steps 5 --warmup 1 --no-extra > $O/prof_$cfg.log 2>&1
done
for f in $(find $O/prof -name '*_results.db' | sort); do python tools/prof_summary.py $f "$(basename $f _results.db): rocprofv3 --kernel-trace --stats -- python bench.py --config $(basename $f _results.db) --steps 5 --warmup 1 --no-extra"; done > $O/kernel_trace_summary.md
pmc() { local cfg=$1 name=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_${cfg}_$name -o $name --output-format csv -- python bench.py --config $cfg --steps 3 --warmup 1 --no-extra > $O/pmc_${cfg}_$name.log 2>&1
  f=$(find $O/pmc_${cfg}_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { echo "## $cfg $name"; python tools/pmc_summary.py "$f"; } >> $O/pmc_summary.txt
}
for cfg in c1 c2 c3; do
  pmc $cfg fetch FETCH_SIZE
  pmc $cfg write WRITE_SIZE
done
pmc c1 sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc c3 sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
# bench lines
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for cfg in c2 c3 c4 c5; do python bench.py --config $cfg --no-extra > $O/bench_$cfg.json 2> $O/bench_$cfg.err; done
python bench.py --config c1 --overlap --no-extra > $O/bench_c1_overlap.json 2>/dev/null
# B sweep at SURVEY 8(d)'s batch sizes (samples per stream and call as in the default shapes)
: > $O/sweep.jsonl
for b in 1 64 4096 16384; do python bench.py --config c1 --batch $b --no-extra 2>/dev/null | tail -1 >> $O/sweep.jsonl; done
for b in 1 16 256 384; do python bench.py --config c2 --batch $b --no-extra 2>/dev/null | tail -1 >> $O/sweep.jsonl; done
for b in 1 4 64 384; do python bench.py --config c3 --batch $b --no-extra 2>/dev/null | tail -1 >> $O/sweep.jsonl; done
find $O -name '*.csv' -size +2M -delete; find $O/prof -type f -size +4M -delete; find $O -name '*.db' -size +4M -delete
ls -la $O | head -50
