#!/bin/bash
# PMC counter passes for the dominant kernel (separate rocprofv3 runs, kernel-trace only).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
CFG=${CFG:-c2}
run() { # name, counters...
  local name=$1; shift
  rm -rf gpurun_out/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_$name -o $name --output-format csv -- python bench.py --config $CFG --steps 3 --warmup 1 --no-extra > gpurun_out/pmc_$name.log 2>&1
  tail -2 gpurun_out/pmc_$name.log | cut -c1-300
  f=$(find gpurun_out/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" | tee gpurun_out/pmc_$name.txt
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
find gpurun_out -name '*.csv' -size +4M -delete
