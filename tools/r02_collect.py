#!/usr/bin/env python3
"""Copies the judged summaries of tools/r02_profile.sh from gpurun_out/r02 (scratch) into profiles/ (tracked) and rebuilds
profiles/pmc_traffic.json (HBM bytes per launch of each workload's front-end kernel: FETCH_SIZE [KiB] x 1024 x 2 -- the gfx950
correction of MI355X_MICROARCH.md, HBM section, which matches the algorithmic byte count of these 8-byte-per-lane streaming reads
to 0.5 % -- plus WRITE_SIZE [KiB] x 1024, separate rocprofv3 --pmc passes)."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r02"), os.path.join(ROOT, "profiles")
for name, out in (("kernel_trace_summary.md", "r02_kernel_trace_summary.md"), ("pmc_summary.txt", "r02_pmc_summary.txt"),
                  ("sweep.jsonl", "r02_batch_sweep.jsonl")):
    shutil.copy(os.path.join(SRC, name), os.path.join(DST, out))
for cfg in ("default", "c2", "c3", "c4", "c5", "c1_overlap"):
    with open(os.path.join(SRC, "bench_%s.json" % cfg)) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    with open(os.path.join(DST, "r02_bench_%s.json" % cfg), "w") as f:
        f.write(lines[-1] + "\n")

pmc = {}
cur = None
for line in open(os.path.join(SRC, "pmc_summary.txt")):
    m = re.match(r"## (c\d) (\w+)", line)
    if m:
        cur = m.groups()
        continue
    m = re.match(r"qrl::(k_decim_\w+)[^ ]* .*?(FETCH_SIZE|WRITE_SIZE)=([0-9.e+]+)", line)
    if m and cur and cur[1] in ("fetch", "write") and "gen" not in m.group(1):
        pmc.setdefault(cur[0], {"kernel": "qrl::" + m.group(1)})["fetch_kib" if m.group(2) == "FETCH_SIZE" else "write_kib"] = float(m.group(3))
out = {}
shape = {"c1": "16384 streams x 262144 samples", "c2": "384 streams x 1638400 samples", "c3": "384 streams x 1638400 samples"}
for cfg, d in pmc.items():
    out[cfg] = {"kernel": d["kernel"], "fetch_bytes": d["fetch_kib"] * 1024 * 2, "write_bytes": d["write_kib"] * 1024,
                "source": "profiles/r02_pmc_summary.txt",
                "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/r02_profile.sh) on `python bench.py --config %s "
                        "--steps 3 --warmup 1 --no-extra`, mean over the launches of the default shape (%s); FETCH_SIZE is reported in KiB and is "
                        "doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)" % (cfg, shape[cfg])}
with open(os.path.join(DST, "pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: (v["fetch_bytes"] + v["write_bytes"]) / 1e9 for k, v in out.items()}))
