#!/bin/bash
# round 3, GPU call 38: two 16-block groups per loop trip at D <= 28 (C2): parity + same-box A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r03al
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 3 $O/pytest.log
for rep in 1 2 3; do
for v in base ng1; do
  L=$PWD/build/libqrl_$v.so; [ $v = base ] && L=$PWD/qradiolink_amd/libqrl_hip.so
  echo "== c2 $v" >> $O/abl.log
  QRL_LIB_PATH=$L python bench.py --config c2 --steps 15 --warmup 3 --no-extra --check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity_check',{}).get('status'))" >> $O/abl.log 2>&1
done
done
cat $O/abl.log
