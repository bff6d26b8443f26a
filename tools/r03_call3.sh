#!/bin/bash
# round 3, GPU call 3: k_decim_pl2 variants (in-flight depth / group size / ring size), fused per-channel kernel k_chan_tail
set -u
export TMPDIR=/tmp
O=gpurun_out/r03c
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chan.py tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --steps 20 --warmup 3 --no-extra --check > $O/bench_c1.json 2> $O/bench_c1.err
for v in pd4g4 pd5g4 pd6g1 rp16pd12 rp16pd14; do
  QRL_LIB_PATH=$PWD/build/libqrl_$v.so python bench.py --steps 20 --warmup 3 --no-extra --check > $O/bench_c1_$v.json 2> $O/bench_c1_$v.err
done
python bench.py --config c4 --steps 20 --warmup 3 --no-extra > $O/bench_c4.json 2> $O/bench_c4.err
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o c4 -- python bench.py --config c4 --steps 5 --warmup 1 --no-extra > $O/prof_c4.log 2>&1
for f in $(find $O/prof -name '*_results.db' | sort); do python tools/prof_summary.py $f "$(basename $f _results.db)"; done > $O/kernel_trace_summary.md 2>&1
find $O -name '*.csv' -size +2M -delete; find $O/prof -type f -size +4M -delete; find $O -name '*.db' -size +4M -delete
for f in $O/bench_c1*.json; do echo $f; python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("parity_check",{}).get("status"))
except Exception as e: print("ERR", e)
P
done
cut -c1-200 $O/bench_c4.json; cat $O/kernel_trace_summary.md
