#!/bin/bash
# quick perf iteration: front-end parity tests, bench summary, phase profile
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q --tb=short --maxfail=5 -m gpu > gpurun_out/tq.log 2>&1; tail -3 gpurun_out/tq.log
timeout 300 python bench.py > gpurun_out/bq.log 2>&1
tail -1 gpurun_out/bq.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], 'C1', d['north_star_c1']['value'], d['north_star_c1']['ms_per_step'], d['north_star_c1']['roofline']['kernel_ms'], d['north_star_c1']['roofline']['frac'])"
(timeout 100 python tools/prof_phases.py c1; timeout 100 python tools/prof_phases.py c2) 2>&1 | grep -v amdgpu
