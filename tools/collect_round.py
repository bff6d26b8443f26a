#!/usr/bin/env python3
"""Copies the judged summaries of tools/profile_round.sh from gpurun_out/<tag> (scratch; tag = argv[1], e.g. r04) into profiles/ (tracked) and rebuilds
profiles/pmc_traffic.json: HBM bytes per launch of each workload's dominant kernel = FETCH_SIZE [KiB] x 1024 / f_fetch +
WRITE_SIZE [KiB] x 1024 / f_write, where f_* are the calibration ratios measured in the same pass on an elementwise kernel of known
size (tools/pmc_calibrate.py; MI355X_MICROARCH.md's gfx950 correction says f_fetch = 0.5), together with the id of the kernel
sources the pass ran on (bench.py reports traffic only for that id)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
SRC, DST = os.path.join(ROOT, "gpurun_out", TAG), os.path.join(ROOT, "profiles")
for name, out in (("kernel_trace_summary.md", TAG + "_kernel_trace_summary.md"), ("pmc_summary.txt", TAG + "_pmc_summary.txt"),
                  ("kernel_alone_summary.md", TAG + "_kernel_alone_summary.md"), ("sweep.jsonl", TAG + "_batch_sweep.jsonl")):
    shutil.copy(os.path.join(SRC, name), os.path.join(DST, out))
bench = {}
for cfg in ("default", "c2", "c3", "c4", "c5", "c1_serial"):
    try:
        with open(os.path.join(SRC, "bench_%s.json" % cfg)) as f:
            lines = [l for l in f.read().splitlines() if l.startswith("{")]
        with open(os.path.join(DST, "%s_bench_%s.json" % (TAG, cfg)), "w") as f:
            f.write(lines[-1] + "\n")
        bench[cfg] = json.loads(lines[-1])
    except (OSError, IndexError):
        print("missing bench line:", cfg)
sid = open(os.path.join(SRC, "source_id.txt")).read().strip()

sections, cur = {}, None
cal = {}
insts = {}
for line in open(os.path.join(SRC, "pmc_summary.txt")):
    m = re.match(r"## (c\d|calibration) (\w+)", line)
    if m:
        cur = m.groups()
        continue
    m = re.search(r"(FETCH_SIZE|WRITE_SIZE) = [0-9.e+]+ KiB .* ratio to 2\^20 KiB = ([0-9.]+)", line)
    if m and cur and cur[0] == "calibration":
        cal[m.group(1)] = float(m.group(2))
        continue
    if cur and cur[1] == "insts" and line.startswith("qrl::"):
        kname = line.split(" SQ_")[0].split(" GRBM_")[0].strip()
        insts.setdefault(cur[0], {})[kname] = {c: (float(v), int(n)) for c, v, n in re.findall(r"(\w+)=([0-9.e+]+)\(n=(\d+)\)", line)}
        continue
    m = re.match(r"(qrl::\S+?)(<.*>)? .*?(FETCH_SIZE|WRITE_SIZE)=([0-9.e+]+)\(n=(\d+)\)", line)
    if m and cur:
        sections.setdefault(cur[0], []).append((m.group(1), m.group(2) or "", m.group(3), float(m.group(4)), int(m.group(5))))
f_fetch, f_write = cal.get("FETCH_SIZE", 0.5), cal.get("WRITE_SIZE", 1.0)
# round the calibration to the documented factors when it is within 3 % of them (the counters' granularity), else keep it
f_fetch = 0.5 if abs(f_fetch - 0.5) < 0.015 else f_fetch
f_write = 1.0 if abs(f_write - 1.0) < 0.03 else f_write
dominant = {"c1": bench.get("default", {}).get("roofline", {}).get("kernel"), "c2": bench.get("c2", {}).get("roofline", {}).get("kernel"),
            "c3": bench.get("c3", {}).get("roofline", {}).get("kernel"), "c4": bench.get("c4", {}).get("roofline", {}).get("kernel"),
            "c5": bench.get("c5", {}).get("roofline", {}).get("kernel")}
out = {"_source_id": sid, "_calibration": {"measured": cal, "applied": {"FETCH_SIZE": f_fetch, "WRITE_SIZE": f_write},
                                           "how": "tools/pmc_calibrate.py under the same rocprofv3 --pmc passes (tools/profile_round.sh): counter / 2^20 KiB on y = x + 1 over 1 GiB"}}
for cfg, rows in sections.items():
    want = (dominant.get(cfg) or "").split(" ")[0].split("<")[0]
    want = want if want.startswith("qrl::") else "qrl::" + want
    best = {}
    for base, targs, cnt, val, n in rows:
        if not base.startswith(want):
            continue
        if cnt not in best or val > best[cnt][0]:
            best[cnt] = (val, base + targs, n)
    if "FETCH_SIZE" in best and "WRITE_SIZE" in best:
        out[cfg] = {"kernel": best["FETCH_SIZE"][1], "fetch_bytes": best["FETCH_SIZE"][0] * 1024 / f_fetch,
                    "write_bytes": best["WRITE_SIZE"][0] * 1024 / f_write, "launches_averaged": best["FETCH_SIZE"][2],
                    "source": "profiles/%s_pmc_summary.txt" % TAG,
                    "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_round.sh) on `python bench.py --config %s "
                            "--steps 3 --warmup 1 --no-extra`, mean over the launches of the default shape; KiB counters, divided by the calibration "
                            "factors above" % cfg}
    else:
        print("no PMC rows for", cfg, want)
    if cfg == "c4" and cfg in out:
        # C4 is a chain of kernels that hand rings to each other: the whole chain's HBM bytes per step = the sum over its kernels (largest
        # mean per kernel name = the default shape), next to the dominant kernel's own
        per = {}
        for base, targs, cnt, val, n in rows:
            k = (base + targs).replace("qrl::", "")
            per.setdefault(k, {})
            per[k][cnt] = max(per[k].get(cnt, 0.0), val)
        byk = {k: v.get("FETCH_SIZE", 0.0) * 1024 / f_fetch + v.get("WRITE_SIZE", 0.0) * 1024 / f_write for k, v in per.items() if "decim_mfma" not in k}
        out[cfg]["chain_bytes"] = sum(byk.values())
        out[cfg]["chain_by_kernel"] = byk
# issue side (C3, C5): wave instructions of ONE receiver call = sum over the RX kernels of (mean per launch x launches) / RX calls, where the
# number of RX calls of the profiled command = launches of the workload's dominant (front-end) kernel
for cfg, ks in insts.items():
    want = (dominant.get(cfg) or "").split(" ")[0].split("<")[0].replace("qrl::", "")
    calls = max([v["SQ_WAVES"][1] for k, v in ks.items() if want and want in k] or [0])
    if not calls:
        print("no instruction counters for", cfg, want)
        continue
    per_call, by_kernel = {}, {}
    for k, v in ks.items():
        if "k_tx_" in k:                       # C5's modulator kernels: not part of a receiver call
            continue
        by_kernel[k] = {c: val * n / calls for c, (val, n) in v.items()}
        for c, x in by_kernel[k].items():
            per_call[c] = per_call.get(c, 0.0) + x
    out[cfg + "_issue"] = {"per_rx_call": per_call, "by_kernel": by_kernel, "rx_calls_profiled": calls, "source": "profiles/%s_pmc_summary.txt" % TAG,
                           "note": "rocprofv3 --kernel-trace --pmc SQ_INSTS_* SQ_WAVES SQ_BUSY_CYCLES (one pass, tools/profile_round.sh) on `python bench.py --config %s "
                                   "--steps 3 --warmup 1 --no-extra`; counters are sums over the 8 XCDs" % cfg}
with open(os.path.join(DST, "pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: round((v["fetch_bytes"] + v["write_bytes"]) / 1e9, 3) for k, v in out.items() if not k.startswith("_") and "fetch_bytes" in v}), out["_calibration"])
print({k: {c: "%.3g" % x for c, x in v["per_rx_call"].items()} for k, v in out.items() if k.endswith("_issue")})
