#!/usr/bin/env python3
"""Per-kernel mean of each PMC counter from a rocprofv3 counter_collection.csv (qrl:: kernels)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r.get("Kernel_Name", "")
        if "qrl::" not in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k, " ".join("%s=%.4g(n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
