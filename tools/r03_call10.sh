#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03j
rm -rf $O; mkdir -p $O
for v in rp8 rp16 rp16abl15; do
  echo "== $v" >> $O/rp.log
  QRL_LIB_PATH=$PWD/build/libqrl_$v.so python bench.py --steps 10 --warmup 2 --no-extra --check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d.get('parity_check',{}).get('status'))" >> $O/rp.log 2>&1
done
cat $O/rp.log
