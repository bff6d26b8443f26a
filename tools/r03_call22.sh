#!/bin/bash
# round 3, GPU call 22: pm kernel software pipelined over the groups (fetch of group g + 1 under the matrix work of group g)
set -u
export TMPDIR=/tmp
O=gpurun_out/r03v
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_lifecycle.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
for c in c1 c2 c3; do
python bench.py --config $c --steps 20 --warmup 5 --no-extra --check > $O/bench_$c.json 2> $O/bench_$c.err
python - "$O/bench_$c.json" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("parity_check",{}).get("status"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
done
for c in c1 c2 c3; do
  echo "== $c phase profile" >> $O/abl.log
  QRL_LIB_PATH=$PWD/build/libqrl_pmprof.so timeout 300 python tools/pm_prof.py $c >> $O/abl.log 2>&1
done
cat $O/abl.log
