#!/bin/bash
# round 3, GPU call 15: where the three-tile phase-major front end spends its time (C2, C3): phase profile + ablations + parity rerun
set -u
export TMPDIR=/tmp
O=gpurun_out/r03o
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_lifecycle.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
for c in c2 c3; do
  echo "== $c phase profile" >> $O/abl.log
  QRL_LIB_PATH=$PWD/build/libqrl_pmprof.so timeout 300 python tools/pm_prof.py $c >> $O/abl.log 2>&1
  for v in base abl1 abl2 abl4 abl8 abl12 abl15; do
    L=$PWD/build/libqrl_$v.so; [ $v = base ] && L=$PWD/qradiolink_amd/libqrl_hip.so
    echo "== $c $v" >> $O/abl.log
    QRL_LIB_PATH=$L python bench.py --config $c --steps 10 --warmup 2 --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'])" >> $O/abl.log 2>&1
  done
done
cat $O/abl.log
