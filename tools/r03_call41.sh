#!/bin/bash
# round 3, GPU call 41: k_pfb_chan64 walking four tiles per workgroup (halo kept in LDS): parity + same-box A/B against the previous kernel
set -u
export TMPDIR=/tmp
O=gpurun_out/r03ao
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chan.py tests/test_gpu_sharding.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for rep in 1 2; do
for v in base pf0; do
  L=$PWD/build/libqrl_$v.so; [ $v = base ] && L=$PWD/qradiolink_amd/libqrl_hip.so
  echo "== c4 $v" >> $O/abl.log
  QRL_LIB_PATH=$L timeout 120 rocprofv3 --kernel-trace --stats -d $O/p_$v -o c4 -- python bench.py --config c4 --steps 6 --warmup 2 --no-extra > $O/run_$v.log 2>&1
  f=$(find $O/p_$v -name '*_results.db' | head -1)
  python tools/prof_summary.py $f $v 2>/dev/null | grep -E "k_chan_tail|k_pfb_chan64|k_symsync" >> $O/abl.log
  grep -o '"ms_per_step": [0-9.]*' $O/run_$v.log | head -1 >> $O/abl.log
  rm -rf $O/p_$v
done
done
cat $O/abl.log
