#!/usr/bin/env python3
"""Turnkey pinning run (needs GNU Radio 3.10 + a checkout of qradiolink; NOT runnable in this repository's container):

    python tools/gr_golden/run_all.py /path/to/qradiolink/src [workdir]

1. builds gr_golden (cmake, tools/gr_golden/CMakeLists.txt) against the reference's own hier-block sources,
2. exports the float16-exact IQ of every committed fixture (tests/golden/*.npz) as raw cf32,
3. runs the REAL reference demodulator (file_source -> make_gr_demod_X -> file_sinks) on each,
4. compares every port with the oracle-minted fixture (compare.py: bits equal, floats within 1e-5 of RMS),
5. writes tests/golden/gr/<name>.npz = the reference's own outputs (all bits, float ports in full) plus the GNU Radio / VOLK
   version strings.  Commit those files: tests/test_golden.py::test_oracle_against_real_reference_fixtures then checks the
   ORACLE against them in every CPU test run and DESIGN.md section 2 can drop "parity unpinned" for the covered modes."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def dry_run(work):
    """--dry-run: everything this script does that needs no GNU Radio -- the case table, the IQ export, the command lines -- so that a CPU test
    keeps the pinning run one command away.  Returns the gr_golden command lines it would run."""
    import make_golden
    iq_dir = os.path.join(work, "iq")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--export-iq", iq_dir])
    cmds = []
    for name, mode, rate, offset, (kind, kw) in make_golden.CASES:
        if rate != 1000000:
            continue
        f = os.path.join(iq_dir, name + ".cf32")
        assert os.path.getsize(f) > 0 and os.path.getsize(f) % 8 == 0, f
        cmds.append([os.path.join(work, "gr_golden"), kind, str(kw.get("sps", 0)), str(kw.get("filter_width", 0)), str(int(kw.get("fm", False))),
                     f, os.path.join(work, "out", name)])
    for fn in ("CMakeLists.txt", "gr_golden.cpp", "compare.py", "arbitrate_ted.py"):
        assert os.path.exists(os.path.join(HERE, fn)), fn
    return cmds


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "--dry-run":
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            for c in dry_run(d):
                print("would run:", " ".join(c))
        return 0
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    src = os.path.abspath(sys.argv[1])
    work = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "build", "gr_golden"))
    os.makedirs(work, exist_ok=True)
    ver = subprocess.run(["gnuradio-config-info", "--version"], capture_output=True, text=True)
    if ver.returncode != 0 or not ver.stdout.strip().startswith("3.10"):
        raise SystemExit("GNU Radio 3.10 is required (gnuradio-config-info --version said %r)" % ver.stdout.strip())
    volk = subprocess.run(["volk-config-info", "--version"], capture_output=True, text=True).stdout.strip()
    subprocess.check_call(["cmake", "-S", HERE, "-B", work, "-DQRADIOLINK_SRC=" + src, "-DCMAKE_BUILD_TYPE=Release"])
    subprocess.check_call(["cmake", "--build", work, "-j"])
    import make_golden
    iq_dir, out_dir = os.path.join(work, "iq"), os.path.join(work, "out")
    os.makedirs(out_dir, exist_ok=True)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--export-iq", iq_dir])
    gr_dir = os.path.join(ROOT, "tests", "golden", "gr")
    os.makedirs(gr_dir, exist_ok=True)
    for name, mode, rate, offset, (kind, kw) in make_golden.CASES:
        if rate != 1000000:
            print("skipping %s: the front end (gr_demod_base) is not a hier block gr_golden can instantiate alone" % name)
            continue
        args = [os.path.join(work, "gr_golden"), kind, str(kw.get("sps", 0)), str(kw.get("filter_width", 0)), str(int(kw.get("fm", False))),
                os.path.join(iq_dir, name + ".cf32"), os.path.join(out_dir, name)]
        print(" ".join(args))
        subprocess.check_call(args)
        ports = {}
        for p, dt in ((0, np.complex64), (1, np.complex64), (2, np.uint8), (3, np.uint8)):
            f = os.path.join(out_dir, "%s.port%d" % (name, p))
            if os.path.exists(f):
                ports["port%d" % p] = np.fromfile(f, dt)
        np.savez_compressed(os.path.join(gr_dir, name + ".npz"), gnuradio=ver.stdout.strip(), volk=volk, **ports)
    rc = subprocess.call([sys.executable, os.path.join(HERE, "compare.py"), out_dir, os.path.join(ROOT, "tests", "golden")])
    print("compare.py exit status", rc, "(0 = the oracle agrees with the real reference on every compared port)")
    return rc


if __name__ == "__main__":
    sys.exit(main())
