#!/bin/bash
# round 3, GPU call 25: whole GPU suite with QRL_OPT_OVERLAP on by default for the 2FSK family; smoke; default bench line
set -u
export TMPDIR=/tmp
O=gpurun_out/r03y
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -n 3 $O/bench_default.err; python - <<'P'
import json
d = json.loads(open("gpurun_out/r03y/bench_default.json").read().strip().splitlines()[-1])
print("C1", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("parity_check", {}).get("status"))
print("  serial", d["roofline"].get("serial_mode"))
for k in ("c2", "c3", "c4", "c5"):
    if k in d: print(k, d[k]["value"], d[k]["ms_per_step"], d[k]["roofline"]["kernel"], d[k]["roofline"]["kernel_ms"], d[k]["roofline"]["frac"])
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("thread_per_block_model", {}).get("value"))
P
