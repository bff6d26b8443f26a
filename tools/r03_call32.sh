#!/bin/bash
# round 3, GPU call 32: which stage of k_chan_tail costs what (developer builds, wrong results), C4 step time
set -u
export TMPDIR=/tmp
O=gpurun_out/r03af
rm -rf $O; mkdir -p $O
for v in base ct1 ct2 ct4 ct8 ct16 ct32 ct63 base; do
  L=$PWD/build/libqrl_$v.so; [ $v = base ] && L=$PWD/qradiolink_amd/libqrl_hip.so
  echo "== c4 $v" >> $O/abl.log
  QRL_LIB_PATH=$L timeout 120 rocprofv3 --kernel-trace --stats -d $O/p_$v -o c4 -- python bench.py --config c4 --steps 6 --warmup 2 --no-extra > $O/run_$v.log 2>&1
  f=$(find $O/p_$v -name '*_results.db' | head -1)
  python tools/prof_summary.py $f $v 2>/dev/null | grep -E "k_chan_tail|k_pfb_chan64|k_symsync" >> $O/abl.log
  tail -1 $O/run_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'])" >> $O/abl.log 2>&1
  rm -rf $O/p_$v
done
cat $O/abl.log
