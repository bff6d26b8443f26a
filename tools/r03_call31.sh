#!/bin/bash
# round 3, GPU call 31: same-box A/B of the interleaved rotator (QRL_PM_SWP) on C2 / C3
set -u
export TMPDIR=/tmp
O=gpurun_out/r03ae
rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
for v in base swp; do
  L=$PWD/build/libqrl_$v.so; [ $v = base ] && L=$PWD/qradiolink_amd/libqrl_hip.so
  for c in c2 c3; do
  echo "== $c $v" >> $O/abl.log
  QRL_LIB_PATH=$L python bench.py --config $c --steps 15 --warmup 3 --no-extra --check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d.get('parity_check',{}).get('status'))" >> $O/abl.log 2>&1
  done
done
done
cat $O/abl.log
