#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03i
rm -rf $O; mkdir -p $O
for v in abl1 abl2 abl4 abl8 abl14 abl15; do
  echo "== $v" >> $O/abl.log
  QRL_LIB_PATH=$PWD/build/libqrl_$v.so python bench.py --steps 10 --warmup 2 --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $O/abl.log 2>&1
done
cat $O/abl.log
