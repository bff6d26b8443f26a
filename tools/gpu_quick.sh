#!/bin/bash
# quick GPU iteration: parity tests + bench (no profiler)
set -u
mkdir -p gpurun_out
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py ${BENCH_ARGS:-} 2>&1 | tail -3 | tee gpurun_out/bench.log
