#!/bin/bash
# round 3, GPU call 27: who stretches the C1 front end in the overlapped mode (developer builds that skip tail kernels: wrong results)
set -u
export TMPDIR=/tmp
O=gpurun_out/r03aa
rm -rf $O; mkdir -p $O
for v in base skip1 skip2 skip3 skip4 skip12 skip15; do
  L=$PWD/build/libqrl_$v.so; [ $v = base ] && L=$PWD/qradiolink_amd/libqrl_hip.so
  echo "== c1 $v" >> $O/abl.log
  QRL_LIB_PATH=$L python bench.py --config c1 --steps 10 --warmup 3 --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $O/abl.log 2>&1
done
timeout 200 rocprofv3 --kernel-trace -d $O/p4 -o c4 -- python bench.py --config c4 --steps 4 --warmup 1 --no-extra > $O/p4.log 2>&1
f=$(find $O/p4 -name '*_results.db' | head -1); python tools/prof_timeline.py $f 300 | grep -v k_decim_mfma | tail -40 > $O/timeline_c4.txt 2>&1
find $O -name '*.db' -delete; find $O -name '*.csv' -size +1M -delete
cat $O/abl.log; cat $O/timeline_c4.txt
