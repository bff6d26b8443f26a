#!/bin/bash
# End-of-round profile pass (run on the GPU box through gpurun): kernel-trace stats of the bench commands of c1 .. c5, HBM-traffic PMC passes of
# their dominant kernels (separate rocprofv3 runs, kernel-trace only, as MI355X_MICROARCH.md prescribes), a FETCH_SIZE / WRITE_SIZE
# calibration on a copy of known size, bench lines.  usage: tools/profile_round.sh r04 -> everything under gpurun_out/r04/;
# tools/collect_round.py r04 turns it into profiles/r04_*.
set -u
export TMPDIR=/tmp
TAG=${1:?round tag, e.g. r04}
O=gpurun_out/$TAG
rm -rf $O; mkdir -p $O
python -c "import bench; print(bench.source_id())" > $O/source_id.txt
for cfg in c1 c2 c3 c4 c5; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o $cfg -- python bench.py --config $cfg --steps 5 --warmup 1 --no-extra > $O/prof_$cfg.log 2>&1
done
for f in $(find $O/prof -name '*_results.db' | sort); do python tools/prof_summary.py $f "$(basename $f _results.db): rocprofv3 --kernel-trace --stats -- python bench.py --config $(basename $f _results.db) --steps 5 --warmup 1 --no-extra"; done > $O/kernel_trace_summary.md
# every kernel ALONE (a sync after every call: nothing of the next call runs beside it): C1, C2, C3 chains and C5's receiver, fused and not
kp() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kprof -o $name -- python tools/kprof.py "$@" > $O/kprof_$name.log 2>&1
}
kp alone_c1 18 16384 262144 1000000 3
kp alone_c2 22 384 1638400 25000000 3
kp alone_c3 26 384 1638400 100000000 3
kp alone_c5rx 26 16384 16384 1000000 4
QRL_KPROF_UNFUSED=1 kp alone_c5rx_unfused 26 16384 16384 1000000 4
for f in $(find $O/kprof -name '*_results.db' | sort); do python tools/prof_summary.py $f "$(basename $f _results.db): rocprofv3 --kernel-trace --stats -- python tools/kprof.py (one call at a time, sync after every call)"; done > $O/kernel_alone_summary.md
pmc() { local cfg=$1 name=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_${cfg}_$name -o $name --output-format csv -- python bench.py --config $cfg --steps 3 --warmup 1 --no-extra > $O/pmc_${cfg}_$name.log 2>&1
  f=$(find $O/pmc_${cfg}_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { echo "## $cfg $name"; python tools/pmc_summary.py "$f"; } >> $O/pmc_summary.txt
}
for cfg in c1 c2 c3 c4 c5; do
  pmc $cfg fetch FETCH_SIZE
  pmc $cfg write WRITE_SIZE
done
# issue-side counters of the two workloads the serial per-stream chain bounds (C3, C5): wave instructions by class, for bench.py's
# roofline.issue (VALU wave-instructions of one RX call / RX-alone time against 1024 SIMDs x one wave64 VALU instruction per 4 cycles)
for cfg in c3 c5; do
  pmc $cfg insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES SQ_BUSY_CYCLES
done
# calibration: a device-to-device copy of 1 GiB (reads 2^20 KiB, writes 2^20 KiB) under the same two counters
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $cnt -d $O/cal_$cnt -o cal --output-format csv -- python tools/pmc_calibrate.py > $O/cal_$cnt.log 2>&1
  f=$(find $O/cal_$cnt -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { echo "## calibration $cnt"; python tools/pmc_calibrate.py --summarize "$f"; } >> $O/pmc_summary.txt
done
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for cfg in c2 c3 c4 c5; do python bench.py --config $cfg --no-extra > $O/bench_$cfg.json 2> $O/bench_$cfg.err; done
python bench.py --config c1 --no-overlap --no-extra > $O/bench_c1_serial.json 2>/dev/null
# B sweep at SURVEY 8(d)'s batch sizes (samples per stream and call as in the default shapes)
: > $O/sweep.jsonl
for b in 1 64 4096 16384; do python bench.py --config c1 --batch $b --no-extra 2>/dev/null | tail -1 >> $O/sweep.jsonl; done
for b in 1 16 256 384; do python bench.py --config c2 --batch $b --no-extra 2>/dev/null | tail -1 >> $O/sweep.jsonl; done
for b in 1 4 64 384 1536; do python bench.py --config c3 --batch $b --no-extra 2>/dev/null | tail -1 >> $O/sweep.jsonl; done
# the whole GPU suite and the smoke entry on the final build
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
find $O -name '*.csv' -size +2M -delete; find $O/prof -type f -size +4M -delete; find $O -name '*.db' -size +4M -delete
cat $O/pmc_summary.txt | cut -c1-220
