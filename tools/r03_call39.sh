#!/bin/bash
# round 3, GPU call 39: soak -- the whole GPU suite three times in a row (flakiness hunt), smoke, a default bench line
set -u
export TMPDIR=/tmp
O=gpurun_out/r03am
rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_$i.log 2>&1; echo "run $i rc $?" >> $O/soak.log; tail -1 $O/pytest_$i.log >> $O/soak.log
done
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/soak.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/soak.log
python - <<'P' >> $O/soak.log
import json
d = json.loads(open("gpurun_out/r03am/bench_default.json").read().strip().splitlines()[-1])
print("C1", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity_check"]["status"], d["source_id"])
for k in ("c2", "c3", "c4", "c5"): print(k, d[k]["value"], d[k]["ms_per_step"], d[k]["roofline"]["frac"], d[k]["roofline"]["traffic"])
P
cat $O/soak.log
