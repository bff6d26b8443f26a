#!/bin/bash
# per-kernel times of the C1 chain (stand-alone, kernels one after another) for the given variant libraries
export TMPDIR=/tmp
rm -rf gpurun_out/prof
for lib in ${VARIANTS:-qradiolink_amd/libqrl_hip.so}; do
  n=$(basename $lib .so)
  QRL_LIB_PATH=$PWD/$lib timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o $n -- python bench.py --config ${CFG:-c1} --steps 5 --warmup 1 --no-extra --no-overlap > gpurun_out/prof_$n.log 2>&1
done
for f in $(find gpurun_out/prof -name "*_results.db"); do python tools/prof_summary.py $f $(basename $f); done | tee gpurun_out/prof_summary_${CFG:-c1}.md
find gpurun_out/prof -type f -size +4M -delete
