#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03m
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_lifecycle.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
python bench.py --steps 20 --warmup 5 --no-extra --check > $O/bench_c1.json 2> $O/bench_c1.err
python bench.py --steps 20 --warmup 5 --no-extra --check --overlap > $O/bench_c1_ovl.json 2> $O/bench_c1_ovl.err
for f in $O/bench_c1.json $O/bench_c1_ovl.json; do python - "$f" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("parity_check",{}).get("status"))
P
done
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o c1ovl -- python bench.py --config c1 --steps 5 --warmup 1 --no-extra --overlap > $O/prof_c1.log 2>&1
for f in $(find $O/prof -name '*_results.db' | sort); do python tools/prof_summary.py $f "$(basename $f _results.db)"; done > $O/kernel_trace_summary.md 2>&1
find $O -name '*.csv' -size +2M -delete; find $O/prof -type f -size +4M -delete; find $O -name '*.db' -size +4M -delete
cat $O/kernel_trace_summary.md
