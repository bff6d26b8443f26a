#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, bench, rocprof kernel trace.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -5 | tee gpurun_out/bench.log
echo "== rocprof"
rm -rf gpurun_out/prof && timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o c2 -- python bench.py --steps 5 --warmup 1 --no-extra > gpurun_out/prof_c2.log 2>&1
tail -3 gpurun_out/prof_c2.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o c1 -- python bench.py --config c1 --steps 5 --warmup 1 --no-extra > gpurun_out/prof_c1.log 2>&1
tail -3 gpurun_out/prof_c1.log
find gpurun_out/prof -type f | head -30
for f in $(find gpurun_out/prof -name '*_results.db'); do python tools/prof_summary.py $f $(basename $f) ; done | tee gpurun_out/prof_summary.md
# keep only small artefacts
find gpurun_out/prof -type f -size +8M -delete
