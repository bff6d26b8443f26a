#!/bin/bash
# round 3, GPU call 40: which phase of k_pfb_chan64 costs what (developer builds, wrong results)
set -u
export TMPDIR=/tmp
O=gpurun_out/r03an
rm -rf $O; mkdir -p $O
for v in pf0 pf1 pf2 pf4 pf8 pf15 pf0; do
  echo "== c4 $v" >> $O/abl.log
  QRL_LIB_PATH=$PWD/build/libqrl_$v.so timeout 120 rocprofv3 --kernel-trace --stats -d $O/p_$v -o c4 -- python bench.py --config c4 --steps 6 --warmup 2 --no-extra > $O/run_$v.log 2>&1
  f=$(find $O/p_$v -name '*_results.db' | head -1)
  python tools/prof_summary.py $f $v 2>/dev/null | grep -E "k_chan_tail|k_pfb_chan64|k_symsync" >> $O/abl.log
  grep -o '"ms_per_step": [0-9.]*' $O/run_$v.log | head -1 >> $O/abl.log
  rm -rf $O/p_$v
done
cat $O/abl.log
