#!/bin/bash
# round 3, GPU call 33: k_chan_tail with scalar tap loads (no LDS tap tables, 5 workgroups per CU): parity + C4 step, same box
set -u
export TMPDIR=/tmp
O=gpurun_out/r03ag
rm -rf $O; mkdir -p $O
for rep in 1 2; do
for v in base st1; do
  L=$PWD/build/libqrl_$v.so; [ $v = base ] && L=$PWD/qradiolink_amd/libqrl_hip.so
  echo "== c4 $v" >> $O/abl.log
  QRL_LIB_PATH=$L timeout 120 rocprofv3 --kernel-trace --stats -d $O/p_$v -o c4 -- python bench.py --config c4 --steps 6 --warmup 2 --no-extra > $O/run_$v.log 2>&1
  f=$(find $O/p_$v -name '*_results.db' | head -1)
  python tools/prof_summary.py $f $v 2>/dev/null | grep -E "k_chan_tail|k_pfb_chan64|k_symsync" >> $O/abl.log
  grep -o '"ms_per_step": [0-9.]*' $O/run_$v.log | head -1 >> $O/abl.log
  rm -rf $O/p_$v
done
done
QRL_LIB_PATH=$PWD/build/libqrl_st1.so timeout 600 python -m pytest tests/test_gpu_chan.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cat $O/abl.log
