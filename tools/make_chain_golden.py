#!/usr/bin/env python3
"""Writes tests/golden/ref/chains.json: the construction log of the reference's hier-block constructors (oracle/_ref/libqrl_rec.so, i.e.
the reference's own gr_demod_*.cpp / gr_mod_*.cpp run against oracle/rec_stub) for every case of tests/test_ref_chains.py.  Needs
/root/reference (make -C oracle ref).  The fixture lets that test run where the reference is absent."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_ref_chains as t  # noqa: E402

cases = t.all_cases()
out = {t.fixture_key(k, a): t.ref_log(k, *a) for k, a in cases}
path = os.path.join(ROOT, "tests", "golden", "ref", "chains.json")
with open(path, "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print("wrote", path, len(out), "logs")
