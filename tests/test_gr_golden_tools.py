"""tools/gr_golden/ -- the one-command pin of the oracle against a real GNU Radio 3.10 -- cannot run in this image (no GNU Radio), but it must
stay runnable: this test imports the two drivers and dry-runs everything in them that needs no GNU Radio."""
import importlib.util
import io
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "gr_golden", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_run_all_dry_run(tmp_path):
    m = _load("run_all")
    cmds = m.dry_run(str(tmp_path))
    assert len(cmds) >= 6 and all(c[0].endswith("gr_golden") and os.path.exists(c[5]) for c in cmds)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gr_golden", "run_all.py"), "--dry-run"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count("would run:") == len(cmds)


def test_arbitrate_ted_self_test():
    """the oracle's own port 1 as the stand-in for GNU Radio's: the contract's candidate (include/qrl_contracts.h: symbol_sync_ff clip(u/2, 1) =
    candidate 0, symbol_sync_cc clip(u, 1) = candidate 2) stays within 1e-5 everywhere, the others leave it (tests/golden/ted_sensitivity.json)"""
    m = _load("arbitrate_ted")
    out = io.StringIO()
    r = m.self_test(out=out)
    assert r[("2fsk1k_1M", 0)] is None and r[("qpsk250k_1M", 2)] is None
    assert r[("2fsk1k_1M", 2)] is not None or r[("2fsk1k_1M", 1)] is not None
    assert r[("qpsk250k_1M", 0)] is not None
    assert "first symbol beyond 1e-5" in out.getvalue()
