"""C++ host side above the C ABI (qradiolink_amd/host/gr_hip_blocks.*): GNU Radio-shaped blocks with the
reference's factory arguments, work() ABI, mailbox ownership and error behaviour.  The driver
tests/host/test_adaptor.cpp calls work() with scheduler-like ragged item counts."""
import os
import subprocess

import numpy as np
import pytest
import torch

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host", "test_adaptor")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])


def test_adaptor_builds_and_fails_loudly_without_device():
    if not os.path.exists(EXE):
        _build()
    r = subprocess.run([EXE, "nodevice"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    if not torch.cuda.is_available():
        # same convention as the reference: device construction throws std::runtime_error (radiocontroller.cpp:1974-1983)
        assert "runtime_error" in r.stdout and "no CPU fallback" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name,fam,sps,fw,fm", [("gmsk10k_1M", "gmsk", 1, 20000, 0), ("gmsk10k_8M", "gmsk", 1, 20000, 0),
                                                ("2fsk1k_1M", "2fsk", 10, 2000, 0), ("qpsk250k_1M", "qpsk", 2, 160000, 0)])
def test_rx_block_matches_golden(tmp_path, name, fam, sps, fw, fm):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    iq = z["iq_f16"].astype(np.float32)
    (tmp_path / "iq.bin").write_bytes(iq.tobytes())
    r = subprocess.run([EXE, "rx", fam, str(sps), str(fw), str(fm), str(int(z["rate"])), str(float(z["offset"])),
                        str(tmp_path / "iq.bin"), str(tmp_path / "a.bin"), str(tmp_path / "b.bin")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    a = np.fromfile(tmp_path / "a.bin", np.uint8)
    want_a = np.unpackbits(z["bits_a"])[: int(z["n_bits_a"])]
    # the mailbox hands bits out in >= 32-bit batches (gr_bit_sink.cpp:48-52): a short tail may still be inside
    assert a.size >= want_a.size - 31 and np.array_equal(a, want_a[: a.size])
    if fam != "qpsk":
        b = np.fromfile(tmp_path / "b.bin", np.uint8)
        want_b = np.unpackbits(z["bits_b"])[: int(z["n_bits_b"])]
        assert b.size >= want_b.size - 31 and np.array_equal(b, want_b[: b.size])


@pytest.mark.gpu
def test_tx_block_matches_oracle(tmp_path):
    data = np.random.default_rng(3).integers(0, 256, 20000, dtype=np.uint8)
    (tmp_path / "bytes.bin").write_bytes(data.tobytes())
    r = subprocess.run([EXE, "tx", str(tmp_path / "bytes.bin"), str(tmp_path / "iq.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "iq.bin", np.complex64)
    ref = orc.mod_qpsk(data)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
