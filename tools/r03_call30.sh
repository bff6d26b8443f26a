#!/bin/bash
# round 3, GPU call 30: FLL LDS window 128 / 64 / 32 / 16 samples in the overlapped C1 step
set -u
export TMPDIR=/tmp
O=gpurun_out/r03ad
rm -rf $O; mkdir -p $O
for rep in 1 2; do
for v in base fll64 fll32 fll16; do
  L=$PWD/build/libqrl_$v.so; [ $v = base ] && L=$PWD/qradiolink_amd/libqrl_hip.so
  for c in "c1" "c1 --no-overlap"; do
  echo "== $c $v" >> $O/abl.log
  QRL_LIB_PATH=$L python bench.py --config $c --steps 10 --warmup 3 --no-extra --check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d.get('parity_check',{}).get('status'))" >> $O/abl.log 2>&1
  done
done
done
cat $O/abl.log
