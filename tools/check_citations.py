#!/usr/bin/env python3
"""Checks that every reference citation `src/...file.ext:LINE[-LINE]` in the docs, header, oracle and kernels names an existing
file of the reference tree and a line range inside it.  Usage: python tools/check_citations.py [/root/reference]
(the reference is only available in the build container; nothing at run time depends on it)."""
import glob
import os
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = [f for pat in ("*.md", "include/*.h", "oracle/*.c", "oracle/*.h", "qradiolink_amd/csrc/*", "qradiolink_amd/host/*", "qradiolink_amd/host/qt/*", "tests/host/*.cpp", "tests/host/*.h", "docs/*.md",
                       "qradiolink_amd/*.py", "tests/*.py", "bench.py") for f in glob.glob(os.path.join(root, pat))]
files = [f for f in files if os.path.isfile(f) and not f.endswith((".o", ".so", ".a")) and os.access(f, os.R_OK) and not (os.access(f, os.X_OK) and "." not in os.path.basename(f))]
files = [f for f in files if os.path.basename(f) not in ("SURVEY.md", "PAPERS.md", "SNIPPETS.md", "BASELINE.md")]
pat = re.compile(r"(?<![\w/])((?:src/)?(?:gr/|DMR/|MMDVM/|M17/)?[A-Za-z_0-9]+\.(?:cpp|h|cc|hpp|pro)):(\d+)(?:-(\d+))?")
lens, bad, n = {}, 0, 0
index = {}
for dp, _, fs in os.walk(ref):
    for f in fs:
        index.setdefault(f, []).append(os.path.join(dp, f))
for path in sorted(files):
    for ln, line in enumerate(open(path, errors="replace"), 1):
        for m in pat.finditer(line):
            name, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            cands = [os.path.join(ref, name)] if os.path.exists(os.path.join(ref, name)) else \
                    [os.path.join(ref, "src", name)] if os.path.exists(os.path.join(ref, "src", name)) else index.get(os.path.basename(name), [])
            if not cands:
                # our own files (engine.cpp:.., kernels) are not reference citations
                if os.path.exists(os.path.join(root, "qradiolink_amd", "csrc", os.path.basename(name))) or "gr-" in line[:m.start()][-40:]:
                    continue
                print("%s:%d: no such reference file: %s" % (os.path.relpath(path, root), ln, name)); bad += 1; continue
            n += 1
            L = lens.setdefault(cands[0], sum(1 for _ in open(cands[0], errors="replace")))
            if not (1 <= a <= b <= L):
                print("%s:%d: %s:%d-%d outside 1..%d" % (os.path.relpath(path, root), ln, name, a, b, L)); bad += 1
print("%d citations checked, %d problems" % (n, bad))
sys.exit(1 if bad else 0)
