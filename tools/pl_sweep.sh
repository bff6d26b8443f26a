#!/bin/bash
# times the C1 front end (overlapped and stand-alone) for every variant library under build/
mkdir -p gpurun_out
for lib in ${VARIANTS:-$(ls build/libqrl_*.so)}; do
  n=$(basename $lib .so)
  a=$(QRL_LIB_PATH=$PWD/$lib timeout 120 python bench.py --config c1 --no-extra --overlap --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])")
  b=$(QRL_LIB_PATH=$PWD/$lib timeout 120 python bench.py --config c1 --no-extra --no-overlap --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])")
  echo "$n ${QRL_PL_SMAX:-} | overlapped(step, kernel, frac): $a | alone: $b"
done | tee -a gpurun_out/pl_sweep.log
