#!/bin/bash
# round 3, GPU call 21: k_dec2_fir with batched loads + interior fast path
set -u
export TMPDIR=/tmp
O=gpurun_out/r03u
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 6 $O/pytest.log
for c in c5 c3; do
python bench.py --config $c --steps 20 --warmup 5 --no-extra --check > $O/bench_$c.json 2> $O/bench_$c.err
python - "$O/bench_$c.json" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("parity_check",{}).get("status"), d.get("config",{}).get("rx_alone_ms_per_step"), d.get("config",{}).get("tx_alone_ms_per_step"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
tail -n 2 $O/bench_$c.err
done
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c5 -- python bench.py --config c5 --steps 5 --warmup 1 --no-extra > $O/prof_c5.log 2>&1
for f in $(find $O -name '*_results.db' | sort); do python tools/prof_summary.py $f "$(basename $f _results.db)"; done > $O/kernel_trace_summary.md 2>&1
find $O -name '*.csv' -size +2M -delete; find $O -type f -path '*prof_*' -size +4M -delete; find $O -name '*.db' -size +4M -delete
cat $O/kernel_trace_summary.md | head -30
