#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small seeded inputs and the ORACLE's outputs for them.

The reference (qradiolink) cannot be built or imported here and ships no vectors, so
these fixtures are minted by oracle/liborc.so (parity unpinned).  Their job is to
(a) freeze the oracle against silent drift and (b) give the GPU tests inputs + expected outputs that
travel to the GPU box.  Re-run this script only when the arithmetic contract changes on purpose:
    python tests/golden/make_golden.py
Inputs are stored as float16-exact complex64 (quantised BEFORE the oracle runs) to keep files small."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc   # noqa: E402
import sig   # noqa: E402

CASES = [
    # name, sig mode, device rate, rx offset, oracle demod kwargs
    ("2fsk1k_1M", "2fsk1k", 1000000, 0.0, ("2fsk", dict(sps=10, filter_width=2000, fm=False))),
    ("2fsk1kfm_1M", "2fsk1kfm", 1000000, 0.0, ("2fsk", dict(sps=10, filter_width=2500, fm=True))),
    ("gmsk10k_1M", "gmsk10k", 1000000, 0.0, ("gmsk", dict(sps=1, filter_width=20000))),
    ("gmsk10k_8M", "gmsk10k", 8000000, 25000.0, ("gmsk", dict(sps=1, filter_width=20000))),   # front end 8:1 = f32-MFMA decimator
    ("qpsk250k_1M", "qpsk250k", 1000000, 0.0, ("qpsk", dict(sps=2, filter_width=160000))),
    ("4fsk2kfm_1M", "4fsk2kfm", 1000000, 0.0, ("4fsk", dict(sps=5, filter_width=3000, fm=True))),
    ("4fsk100k_1M", "4fsk100k", 1000000, 0.0, ("4fsk", dict(sps=2, filter_width=125000, fm=True))),
    ("bpsk2k_1M", "bpsk2k", 1000000, 0.0, ("bpsk", dict(sps=5))),
]
# round 6: the same chains behind SURVEY.md 8(d)'s channel (sig.SPEC: 0.37-sample fractional delay, +20 ppm clock error, Es/N0 12 dB) --
# the inputs under which the timing loops' acquisition matters most, so the first GNU Radio run (tools/gr_golden/run_all.py) pins them too
CASES_8D = [
    ("2fsk1k_1M_8d", "2fsk1k", 1000000, 0.0, ("2fsk", dict(sps=10, filter_width=2000, fm=False))),
    ("gmsk10k_1M_8d", "gmsk10k", 1000000, 0.0, ("gmsk", dict(sps=1, filter_width=20000))),
    ("qpsk250k_1M_8d", "qpsk250k", 1000000, 0.0, ("qpsk", dict(sps=2, filter_width=160000))),
    ("4fsk2kfm_1M_8d", "4fsk2kfm", 1000000, 0.0, ("4fsk", dict(sps=5, filter_width=3000, fm=True))),
    ("bpsk2k_1M_8d", "bpsk2k", 1000000, 0.0, ("bpsk", dict(sps=5))),
]
CASES = CASES + CASES_8D


def quantise(x):
    v = x.view(np.float32).astype(np.float16).astype(np.float32)
    return v.view(np.complex64)


def run_oracle(kind, kw, x, rate, offset):
    fe = orc.frontend(x, rate, offset)
    return {"2fsk": orc.demod_2fsk, "gmsk": orc.demod_gmsk, "qpsk": orc.demod_qpsk, "4fsk": orc.demod_4fsk,
            "bpsk": orc.demod_bpsk}[kind](fe, **kw)


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--export-iq":   # raw cf32 inputs of the committed fixtures (for tools/gr_golden)
        os.makedirs(sys.argv[2], exist_ok=True)
        for name, *_ in CASES:
            z = np.load(os.path.join(HERE, name + ".npz"))
            z["iq_f16"].astype(np.float32).tofile(os.path.join(sys.argv[2], name + ".cf32"))
        return
    only = set(sys.argv[1:])   # optional: names of the fixtures to (re)generate; default = all
    for name, mode, rate, offset, (kind, kw) in CASES:
        if only and name not in only:
            continue
        nframes = 1 if rate > 1000000 else (3 if mode.startswith("2fsk") else 4 if mode.startswith("bpsk") else 2)
        y, payloads = sig.make_stream(mode, nframes=nframes, device_rate=rate, rx_offset_hz=offset, seed=77, amp=0.25,
                                      impair=sig.SPEC if name.endswith("_8d") else None)
        if mode.startswith(("4fsk", "bpsk")):
            y = np.concatenate([y, np.zeros(20000, np.complex64)])   # flush the long RRC filters / Viterbi frames
        y = quantise(y[: y.size & ~1])
        r = run_oracle(kind, kw, y, rate, offset)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            iq_f16=y.view(np.float32).astype(np.float16), rate=rate, offset=offset,
            bits_a=np.packbits(r["bits_a"]), n_bits_a=r["bits_a"].size,
            bits_b=np.packbits(r["bits_b"]), n_bits_b=r["bits_b"].size,
            filtered_sha256=digest(r["filtered"]), constellation_sha256=digest(r["constellation"]),
            n_filtered=r["filtered"].size, n_constellation=r["constellation"].size,
            filtered_head=r["filtered"][:64], constellation_head=r["constellation"][:64],
            payloads=np.frombuffer(b"".join(payloads), np.uint8), payload_len=len(payloads[0]))
        print(name, y.size, "samples ->", r["bits_a"].size, "bits", os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
