#!/bin/bash
# round 3, GPU call 36: SQ occupancy / wait counters of the front-end kernel on C1, C2, C3 (separate PMC passes, kernel-trace only)
set -u
export TMPDIR=/tmp
O=gpurun_out/r03aj
rm -rf $O; mkdir -p $O
for cfg in c1 c2 c3; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/sq_$cfg -o sq --output-format csv -- python bench.py --config $cfg --steps 3 --warmup 1 --no-extra --no-overlap > $O/sq_$cfg.log 2>&1
  f=$(find $O/sq_$cfg -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { echo "## $cfg (serial order: the front-end kernel alone on the chip)"; python tools/pmc_summary.py "$f" | grep -E "k_decim_pm|k_fll|k_2fsk_ff|k_qpsk_pipe4|k_fec"; } >> $O/pmc_sq_summary.txt
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/mf_$cfg -o mf --output-format csv -- python bench.py --config $cfg --steps 3 --warmup 1 --no-extra --no-overlap > $O/mf_$cfg.log 2>&1
  f=$(find $O/mf_$cfg -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { echo "## $cfg matrix / LDS counters"; python tools/pmc_summary.py "$f" | grep -E "k_decim_pm"; } >> $O/pmc_sq_summary.txt
done
find $O -name '*.csv' -size +1M -delete; find $O -name '*.db' -delete
cat $O/pmc_sq_summary.txt | cut -c1-400; tail -3 $O/mf_c1.log
