#!/bin/bash
# C4 steady-state timeline under rocprofv3 --kernel-trace: tools/c4_tl.sh OUT name [ENV=VAL ...]
export TMPDIR=/tmp
O=gpurun_out/$1; name=$2; shift 2
mkdir -p $O
env "$@" timeout 300 rocprofv3 --kernel-trace -d $O/tl_$name -o c4 -- python bench.py --config c4 --steps ${QRL_TL_BENCH_STEPS:-8} --warmup 2 --no-extra ${QRL_TL_ARGS:-} > $O/tl_$name.log 2>&1
f=$(find $O/tl_$name -name 'c4_results.db' | head -1)
{ echo "## $name ($*): rocprofv3 --kernel-trace -- python bench.py --config c4 --steps ${QRL_TL_BENCH_STEPS:-8} --warmup 2 --no-extra"; python tools/c4_timeline.py $f ${QRL_TL_STEPS:-4}; } | tee $O/timeline_$name.log
rm -rf $O/tl_$name
