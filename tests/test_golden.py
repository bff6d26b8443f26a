"""Golden fixtures (tests/golden/*.npz, minted by tests/golden/make_golden.py from the oracle):
CPU: the oracle still reproduces them and the transmitted frames are in the decoded bits;
GPU: the HIP path reproduces them through the C ABI (bits exact, float ports by SHA-256)."""
import glob
import hashlib
import os

import numpy as np
import pytest

import orc
import sig

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
ORACLE = {"2fsk1k": ("2fsk", dict(sps=10, filter_width=2000, fm=False)), "2fsk1kfm": ("2fsk", dict(sps=10, filter_width=2500, fm=True)),
          "gmsk10k": ("gmsk", dict(sps=1, filter_width=20000)), "qpsk250k": ("qpsk", dict(sps=2, filter_width=160000)),
          "4fsk2kfm": ("4fsk", dict(sps=5, filter_width=3000, fm=True)), "4fsk100k": ("4fsk", dict(sps=2, filter_width=125000, fm=True)),
          "bpsk2k": ("bpsk", dict(sps=5))}
MODEM = {"2fsk1k": 18, "2fsk1kfm": 16, "gmsk10k": 22, "qpsk250k": 26, "4fsk2kfm": 5, "4fsk100k": 27, "bpsk2k": 0}
DEMOD = {"2fsk": orc.demod_2fsk, "gmsk": orc.demod_gmsk, "qpsk": orc.demod_qpsk, "4fsk": orc.demod_4fsk, "bpsk": orc.demod_bpsk}
# sync word, frame bits, frames that may be lost to acquisition (BPSK: agc2 + FLL + Costas settle during the first frame)
FRAMING = {"gmsk10k": (bytes([0xED, 0x89]), 384, 0), "qpsk250k": (bytes([0xDE, 0x98, 0xAA]), 1516 * 8, 0),
           "4fsk2kfm": (bytes([0xED, 0x89, 0xAA]), 56, 0), "4fsk100k": (bytes([0xDE, 0x98, 0xAA]), 1516 * 8, 0),
           "bpsk2k": (bytes([0xED, 0x89, 0xAA]), 56, 1)}
SINGLE_BRANCH = ("qpsk250k", "4fsk2kfm", "4fsk100k")


def _load(path):
    z = np.load(path)
    iq = z["iq_f16"].astype(np.float32).view(np.complex64)
    bits_a = np.unpackbits(z["bits_a"])[: int(z["n_bits_a"])]
    bits_b = np.unpackbits(z["bits_b"])[: int(z["n_bits_b"])]
    return z, iq, bits_a, bits_b


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_fixtures_exist():
    assert len(FILES) >= 8


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(path):
    z, iq, bits_a, bits_b = _load(path)
    mode = os.path.basename(path).split("_")[0]
    kind, kw = ORACLE[mode]
    fe = orc.frontend(iq, int(z["rate"]), float(z["offset"]))
    r = DEMOD[kind](fe, **kw)
    assert np.array_equal(r["bits_a"], bits_a) and np.array_equal(r["bits_b"], bits_b)
    assert _sha(r["filtered"]) == str(z["filtered_sha256"]) and _sha(r["constellation"]) == str(z["constellation_sha256"])
    # the frames that were transmitted are in the decoded bits (gr_modem::findSync-style search)
    plen = int(z["payload_len"])
    payloads = [bytes(z["payloads"][i:i + plen]) for i in range(0, z["payloads"].size, plen)]
    sync, nbits, may_lose = FRAMING.get(mode, (bytes([0xB5]), 32, 0))
    found = 0
    for bits in (bits_a, bits_b):
        for inv in ((0, 1) if mode.startswith("bpsk") else (0,)):   # BPSK: 180 degree ambiguity, the code is inversion transparent
            fr = sig.find_frames(bits ^ inv, sync, nbits)
            found = max(found, sum(((bytes([0xAA]) + p) if mode == "gmsk10k" else p) in fr for p in payloads))
    if os.path.basename(path).endswith("_8d.npz"):
        # SURVEY 8(d)'s channel (sig.SPEC): at Es/N0 12 dB per channel symbol with a sliding symbol phase the REFERENCE chain itself loses
        # frames during acquisition (2FSK-1k keeps one of three, BPSK-2k two of four; at 16 dB all of them: measured with the oracle) --
        # the fixture freezes what the chain does there, decoded frames are only a sanity floor
        assert found >= 1
        return
    assert found >= len(payloads) - may_lose


@pytest.mark.gpu
@pytest.mark.parametrize("path", [f for f in FILES if os.path.basename(f).split("_")[0] in MODEM],
                         ids=[os.path.basename(f)[:-4] for f in FILES if os.path.basename(f).split("_")[0] in MODEM])
@pytest.mark.parametrize("chunk", [1 << 22, 30000])
def test_hip_reproduces_golden(qrl_ctx, path, chunk):
    import torch
    import qradiolink_amd as q
    z, iq, bits_a, bits_b = _load(path)
    mode = os.path.basename(path).split("_")[0]
    dem = q.Demod(qrl_ctx, MODEM[mode], batch=2, max_chunk=chunk, device_samp_rate=int(z["rate"]), carrier_offset_hz=float(z["offset"]))
    out = q.collect(dem, torch.from_numpy(np.stack([iq, iq])).cuda(), chunk)
    dem.close()
    for b in range(2):
        assert np.array_equal(out["bits_a"][b], bits_a)
        if mode not in SINGLE_BRANCH:
            assert np.array_equal(out["bits_b"][b], bits_b)
        assert _sha(out["filtered"][b].astype(np.complex64)) == str(z["filtered_sha256"])
        assert _sha(out["constellation"][b].astype(np.complex64)) == str(z["constellation_sha256"])


def test_oracle_against_real_reference_fixtures():
    """tests/golden/gr/*.npz = outputs of the REAL reference flowgraph (GNU Radio 3.10 + qradiolink hier blocks) produced by
    tools/gr_golden/run_all.py on a machine that has them.  None are committed yet (parity unpinned): the test
    skips; the day they exist it pins the oracle: hard bits equal, float ports within 1e-5 of RMS."""
    import glob
    GOLD = os.path.join(HERE, "golden")
    files = sorted(glob.glob(os.path.join(GOLD, "gr", "*.npz")))
    if not files:
        pytest.skip("no real-reference fixtures under tests/golden/gr (run tools/gr_golden/run_all.py where GNU Radio 3.10 exists)")
    import sys
    sys.path.insert(0, GOLD)
    import make_golden
    cases = {c[0]: c for c in make_golden.CASES}
    for path in files:
        name = os.path.basename(path)[:-4]
        z, g = np.load(os.path.join(GOLD, name + ".npz")), np.load(path)
        _, mode, rate, offset, (kind, kw) = cases[name]
        x = z["iq_f16"].astype(np.float32).view(np.complex64)
        r = make_golden.run_oracle(kind, kw, x, int(z["rate"]), float(z["offset"]))
        for port, key in (("port2", "bits_a"), ("port3", "bits_b")):
            if port in g and r[key].size:
                n = min(g[port].size, r[key].size)
                assert n > 0 and np.array_equal(g[port][:n], r[key][:n]), "%s %s differs from the real reference" % (name, key)
        for port, key in (("port0", "filtered"), ("port1", "constellation")):
            if port in g and r[key].size:
                n = min(g[port].size, r[key].size)
                rms = np.sqrt(np.mean(np.abs(g[port][:n]) ** 2)) + 1e-30
                assert np.max(np.abs(g[port][:n] - r[key][:n])) / rms <= 1e-5, "%s %s above 1e-5 of RMS" % (name, key)


def test_oracle_against_stock_block_fixtures():
    """tests/golden/gr/stock_<name>.npy = what GNU Radio 3.10 itself returned for the block-level cases of tools/gr_golden/stock_blocks.py (the firdes designers the
    chains and their setters call, analog::sig_source_f).  None are committed yet -- the test skips; the day they exist the oracle must equal them."""
    import glob
    import importlib.util
    files = sorted(glob.glob(os.path.join(HERE, "golden", "gr", "stock_*.npy")))
    if not files:
        pytest.skip("no stock-block fixtures under tests/golden/gr (run tools/gr_golden/stock_blocks.py where GNU Radio 3.10 exists)")
    spec = importlib.util.spec_from_file_location("stock_blocks", os.path.join(os.path.dirname(HERE), "tools", "gr_golden", "stock_blocks.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    table = m.cases()
    for path in files:
        name = os.path.basename(path)[len("stock_"):-4]
        assert name in table, name
        msg = m.compare(name, np.load(path), table[name][1]())
        assert msg is None, msg

