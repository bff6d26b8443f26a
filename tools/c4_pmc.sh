#!/bin/bash
# SQ counters of the C4 kernels, every kernel alone (tools/c4_alone.py): tools/c4_pmc.sh OUT
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
run() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$name -o $name --output-format csv -- python tools/c4_alone.py 4 > $O/pmc_$name.log 2>&1
  f=$(find $O/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { echo "## $name: $*"; python tools/pmc_summary.py "$f"; } >> $O/c4_pmc.txt
  rm -rf $O/pmc_$name
}
run sq GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
run lds SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM
cat $O/c4_pmc.txt | cut -c1-400
