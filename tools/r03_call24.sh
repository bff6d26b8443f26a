#!/bin/bash
# round 3, GPU call 24: C1 modes on one box: serial, overlapped, overlapped + slim FLL workgroups
set -u
export TMPDIR=/tmp
O=gpurun_out/r03x
rm -rf $O; mkdir -p $O
for rep in 1 2; do
for m in "" "--overlap" "--overlap --fll-slim" "--fll-slim"; do
  echo "== c1 $m" >> $O/abl.log
  python bench.py --config c1 --steps 15 --warmup 3 --no-extra --check $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d.get('parity_check',{}).get('status'))" >> $O/abl.log 2>&1
done
done
cat $O/abl.log
