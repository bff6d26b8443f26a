"""The asm-issued tile prefetch of k_decim_mfma (global loads into accumulator registers behind the compiler's back, explicit
s_waitcnt vmcnt(0) later) is only sound if nothing touches those registers between issue and wait.  tools/audit_mfma_prefetch.py
walks every control-flow path of every instantiation in the generated gfx950 assembly; this test keeps that proof in the suite
(hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_instruction_touches_a_prefetch_register_before_its_wait():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_mfma_prefetch.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "0 problems" in r.stdout
