#!/bin/bash
# Final-profile pass of a round: kernel-trace stats for both bench workloads + HBM traffic PMC passes
# (separate rocprofv3 runs, kernel-trace only, as the MI355X guide prescribes).  Everything under gpurun_out/.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof gpurun_out/pmc_*
for cfg in c2 c1; do
  timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o $cfg -- python bench.py --config $cfg --steps 5 --warmup 1 --no-extra > gpurun_out/prof_$cfg.log 2>&1
done
for f in $(find gpurun_out/prof -name '*_results.db'); do python tools/prof_summary.py $f $(basename $f); done | tee gpurun_out/prof_summary.md
run() { local cfg=$1 name=$2; shift 2
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_${cfg}_$name -o $name --output-format csv -- python bench.py --config $cfg --steps 3 --warmup 1 --no-extra > gpurun_out/pmc_${cfg}_$name.log 2>&1
  f=$(find gpurun_out/pmc_${cfg}_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { echo "## $cfg $name"; python tools/pmc_summary.py "$f"; } | tee -a gpurun_out/pmc_summary.txt
}
rm -f gpurun_out/pmc_summary.txt
for cfg in c2 c1; do
  run $cfg fetch FETCH_SIZE
  run $cfg write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
done
run c2 sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
find gpurun_out -name '*.csv' -size +2M -delete; find gpurun_out/prof -type f -size +4M -delete
