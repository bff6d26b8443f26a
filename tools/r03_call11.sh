#!/bin/bash
# round 3, GPU call 11: whole GPU suite after the pm DPP change + all-to-all emulation test; default bench line (c1 + c2..c5 sub-lines)
set -u
export TMPDIR=/tmp
O=gpurun_out/r03k
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 6 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -n 3 $O/bench_default.err; python - <<'P'
import json
d = json.loads(open("gpurun_out/r03k/bench_default.json").read().strip().splitlines()[-1])
print("C1", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("parity_check", {}).get("status"))
print("  overlap", d["roofline"].get("overlapped_mode"))
for k in ("c2", "c3", "c4", "c5"):
    if k in d: print(k, d[k]["value"], d[k]["ms_per_step"], d[k]["roofline"]["kernel"], d[k]["roofline"]["kernel_ms"], d[k]["roofline"]["frac"])
if "c4" in d and "freq_xlating_form" in d["c4"]: print("c4 form2", d["c4"]["freq_xlating_form"]["value"], d["c4"]["freq_xlating_form"]["roofline"]["note"])
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
P
