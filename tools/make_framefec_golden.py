"""Regenerates tests/golden/ref/framefec.npz from the REAL reference (oracle/_ref/libqrl_ref.so = /root/reference's BPTC19696.cpp,
Hamming.cpp, M17FrameDecoder.cpp, M17FrameEncoder.cpp, M17Viterbi.hpp, M17Golay.cpp compiled where they lie: make -C oracle ref).
Runs only in the build container (needs /root/reference); the vectors it writes travel with the repository."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libqrl_ref.so"))
P = lambda a: a.ctypes.data_as(C.c_void_p)
rng = np.random.default_rng(20260926)


def flip(x, k, lo=0):
    x = x.copy()
    for _ in range(k):
        j = int(rng.integers(lo, x.size * 8))
        x[j >> 3] ^= 1 << (j & 7)
    return x


# ---- BPTC(196,96)
n = 600
pay = rng.integers(0, 256, (n, 12), dtype=np.uint8)
base = rng.integers(0, 256, (n, 33), dtype=np.uint8)
enc = base.copy()
for i in range(n):
    ref.ref_bptc19696_encode(P(pay[i]), P(enc[i]))
rx = np.stack([enc[i] if i % 5 == 0 else rng.integers(0, 256, 33, dtype=np.uint8) if i % 5 == 4 else flip(enc[i], int(rng.integers(1, 10))) for i in range(n)])
dec = np.zeros((n, 12), np.uint8)
for i in range(n):
    ref.ref_bptc19696_decode(P(rx[i]), P(dec[i]))

# ---- M17
frames, types, lsfs, streams = [], [], [], []
seq_frames, seq_lsf = [], []
for t in range(120):
    lsf = rng.integers(0, 256, 28, dtype=np.uint8)
    pl = rng.integers(0, 256, (7, 16), dtype=np.uint8)
    fr = np.zeros((8, 48), np.uint8)
    ref.ref_m17_encode(P(lsf), P(pl), 7, P(fr))
    if t < 12:   # LSF reassembly from the LICH segments: six stream frames through one decoder, no LSF frame
        s = np.ascontiguousarray(fr[1:7]) if t % 2 == 0 else np.ascontiguousarray(np.stack([flip(f, 3, lo=16) for f in fr[1:7]]))
        a, b = np.zeros(30, np.uint8), np.zeros(18, np.uint8)
        ref.ref_m17_decode_sequence(P(s), 6, P(a), P(b))
        seq_frames.append(s); seq_lsf.append(a)
    for i in range(8):
        f = fr[i]
        m = (t + i) % 4
        if m == 1:
            f = flip(f, int(rng.integers(1, 12)), lo=16)
        elif m == 2:
            f = flip(f, int(rng.integers(1, 40)))
        elif m == 3 and i % 3 == 0:
            f = rng.integers(0, 256, 48, dtype=np.uint8); f[:2] = fr[i][:2]
        a, b = np.zeros(30, np.uint8), np.zeros(18, np.uint8)
        ty = ref.ref_m17_decode_frame(P(np.ascontiguousarray(f)), P(a), P(b))
        frames.append(f.copy()); types.append(ty); lsfs.append(a); streams.append(b)

# ---- M17 encoder: the reference's M17FrameEncoder (stateful) against stateless records {type, lich ok, payload[30], LICH segment[6], 0, 0}
enc_rec, enc_frames = [], []
for t in range(20):
    lsf28 = rng.integers(0, 256, 28, dtype=np.uint8)
    pl = rng.integers(0, 256, (9, 16), dtype=np.uint8)
    fr = np.zeros((10, 48), np.uint8)
    ref.ref_m17_encode(P(lsf28), P(pl), 9, P(fr))
    a, b = np.zeros(30, np.uint8), np.zeros(18, np.uint8)
    ref.ref_m17_decode_frame(P(np.ascontiguousarray(fr[0])), P(a), P(b))      # the LSF with the CRC the reference computed
    rec = np.zeros(40, np.uint8); rec[0] = 1; rec[2:32] = a
    enc_rec.append(rec); enc_frames.append(fr[0].copy())
    for i in range(9):
        fn = i | (0x8000 if i == 8 else 0)
        rec = np.zeros(40, np.uint8); rec[0] = 2; rec[1] = 1; rec[2] = fn >> 8; rec[3] = fn & 0xFF; rec[4:20] = pl[i]
        k = i % 6
        rec[32:37] = a[5 * k:5 * k + 5]; rec[37] = k
        enc_rec.append(rec); enc_frames.append(fr[i + 1].copy())

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref", "framefec.npz"),
                    bptc_payload=pay, bptc_base=base, bptc_encoded=enc, bptc_rx=rx, bptc_decoded=dec,
                    m17_frames=np.stack(frames), m17_type=np.array(types, np.uint8), m17_lsf=np.stack(lsfs), m17_stream=np.stack(streams),
                    m17_seq_frames=np.stack(seq_frames), m17_seq_lsf=np.stack(seq_lsf),
                    m17_enc_records=np.stack(enc_rec), m17_enc_frames=np.stack(enc_frames))
print("wrote", n, "bursts,", len(frames), "M17 frames,", len(seq_frames), "LICH sequences")
