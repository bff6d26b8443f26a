#!/bin/bash
# PMC passes for the C1 front end (stand-alone): SQ wait/issue breakdown, HBM traffic.  Separate rocprofv3 runs, kernel-trace only.
export TMPDIR=/tmp
mkdir -p gpurun_out
CFG=${CFG:-c1}
run() { local name=$1; shift
  rm -rf gpurun_out/pmc_${CFG}_$name
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_${CFG}_$name -o $name --output-format csv -- python bench.py --config $CFG --steps 3 --warmup 1 --no-extra ${BENCH_EXTRA:---no-overlap} > gpurun_out/pmc_${CFG}_$name.log 2>&1
  f=$(find gpurun_out/pmc_${CFG}_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { echo "## $CFG $name ${QRL_LIB_PATH:-}"; python tools/pmc_summary.py "$f"; } | tee -a gpurun_out/pmc_summary_${CFG}.txt
}
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
find gpurun_out -name '*.csv' -size +2M -delete
