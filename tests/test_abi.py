"""C-ABI boundary (include/qrl_hip.h) on CPU: the library loads, exports every declared symbol,
its host-side filter design / tables agree bit-for-bit with the independently written oracle, and
the device path fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import orc
import qradiolink_amd as q

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAS_GPU = torch.cuda.is_available()


def _declared():
    src = open(os.path.join(ROOT, "include", "qrl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(qrl_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = q.load_library()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libqrl_hip.so does not export %s" % n
    assert sorted(q.EXPORTED_SYMBOLS) == names
    assert b"gfx950" in lib.qrl_version()


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C with no HIP/torch types."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "qrl_hip.h"\nint main(void){qrl_demod_config c; (void)c; return QRL_OK;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", src,
                               "-o", os.path.join(d, "t.o")])


def test_strerror_covers_status_codes():
    lib = q.load_library()
    seen = {lib.qrl_strerror(c) for c in range(0, -7, -1)}
    assert len(seen) == 7 and b"unknown" not in seen
    assert lib.qrl_strerror(-99) == b"unknown"


@pytest.mark.parametrize("args", [(1, 1e6, 10e3, 10e3, 5), (1, 25e6, 480e3, 100e3, 5), (2, 2e6, 40e3, 40e3, 5),
                                  (1, 20e3, 2e3, 2e3, 0), (1, 80e3, 20e3, 20e3, 0)])
def test_host_low_pass_equals_oracle(args):
    assert np.array_equal(q.low_pass(*args).view(np.uint32), orc.low_pass(*args).view(np.uint32))


def test_host_other_designs_equal_oracle():
    a = q.low_pass_2(1, 1e6, 250e3, 50e3, 60, 5)
    assert np.array_equal(a.view(np.uint32), orc.low_pass_2(1, 1e6, 250e3, 50e3, 60, 5).view(np.uint32))
    b = q.complex_band_pass(1, 20e3, -2e3, 0, 2e3, 5)
    assert np.array_equal(b.view(np.uint32), orc.complex_band_pass(1, 20e3, -2e3, 0, 2e3, 5).view(np.uint32))
    c = q.root_raised_cosine(1, 20000, 2000, 0.2, 351)
    assert np.array_equal(c.view(np.uint32), orc.root_raised_cosine(1, 20000, 2000, 0.2, 351).view(np.uint32))
    for name, n in (("mmse", 129 * 8), ("atan", 257), ("tanh", 256)):
        assert np.array_equal(q.table(name).view(np.uint32), orc.table(name, n).view(np.uint32)), name
    for hz, fs in ((25000.0, 25e6), (-1200.0, 1e6), (0.0, 1e6)):
        r = 2 * np.pi * -hz / fs
        assert q.load_library().qrl_phase_inc_to_turn(r) == orc.phase_inc_to_turn(r)


def test_null_arguments_are_rejected_not_crashed():
    lib = q.load_library()
    assert lib.qrl_init(0, None) == -1
    assert lib.qrl_demod_create(None, None, None) == -1
    assert lib.qrl_demod_process(None, None, 0, 0, None) == -1
    assert lib.qrl_demod_sync(None) == -1
    assert lib.qrl_demod_reset(None) == -1
    lib.qrl_demod_destroy(None)
    lib.qrl_shutdown(None)


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_no_cpu_fallback_without_device():
    lib = q.load_library()
    h = C.c_void_p()
    assert lib.qrl_init(0, C.byref(h)) == -2          # QRL_ERR_NO_DEVICE
    assert not h.value
    with pytest.raises(q.QrlError, match="no usable HIP device"):
        q.Context(0)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(q, "_lib", None)
    monkeypatch.setattr(q, "LIB_PATH", str(tmp_path / "libqrl_hip.so"))
    with pytest.raises(q.QrlError, match="no CPU fallback"):
        q.load_library()


def test_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under qradiolink_amd/ may include, link, load or call it
    (comments may cite the arithmetic contract it documents)."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "qradiolink_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"liborc|#include\s*[<\"][^>\"]*orc|import orc|from orc|orc_[a-z0-9_]+\s*\(|-lorc", text):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
