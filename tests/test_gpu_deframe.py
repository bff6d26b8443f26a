"""GPU parity of the device deframer (gr_deframer_bb) against the oracle: random bit streams with sync words planted at
random places, ragged per-stream counts, state carried across calls, and the demodulator -> deframer chain."""
import numpy as np
import pytest

import orc
import sig

pytestmark = pytest.mark.gpu

SYNC = {1: [0x89ED, 0xED89, 0x98DE, 0xED77, 0x8CC8, 0x4C8A2B], 2: [0xB5, 0x4C8A2B], 3: [0x89ED, 0xED89, 0x4C8A2B]}


def _stream(rng, n, type_):
    bits = rng.integers(0, 2, n, dtype=np.uint8)
    pos = 5
    while pos + 40 < n:
        w = SYNC[type_][rng.integers(0, len(SYNC[type_]))]
        nb = 24 if w > 0xFFFF else (8 if w < 0x100 else 16)
        bits[pos:pos + nb] = [(w >> (nb - 1 - k)) & 1 for k in range(nb)]
        pos += int(rng.integers(30, 700))
    return bits


@pytest.mark.parametrize("type_", [1, 2, 3])
@pytest.mark.parametrize("cuts", [[4000], [1, 63, 64, 65, 1000, 7, 2800], [129] * 31])
def test_deframer_bit_exact(qrl_ctx, type_, cuts):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(type_ * 100 + len(cuts))
    B, total = 5, sum(cuts)
    data = np.stack([_stream(rng, total, type_) for _ in range(B)])
    d = torch.from_numpy(data).cuda()
    df = q.Deframer(qrl_ctx, type_, B)
    got = [[] for _ in range(B)]
    states = [np.zeros(3, np.uint32) for _ in range(B)]
    want = [[] for _ in range(B)]
    pos = 0
    for c in cuts:
        # ragged: stream b only has c - (b % 3) valid bits in this call (the rest are skipped, like a short demod output)
        counts = np.array([max(c - (b % 3), 0) for b in range(B)], np.int32)
        out, oc = df.process(d[:, pos:pos + c].contiguous(), counts=torch.from_numpy(counts).cuda(), count_stride=1, n=c)
        out, oc = out.cpu().numpy(), oc.cpu().numpy()
        for b in range(B):
            got[b].append(out[b, :oc[b]].copy())
            want[b].append(orc.deframer(type_, data[b, pos:pos + counts[b]], states[b]))
        pos += c
    df.close()
    nonempty = 0
    for b in range(B):
        g, w = np.concatenate(got[b]), np.concatenate(want[b])
        assert g.size == w.size and np.array_equal(g, w), "stream %d" % b
        nonempty += w.size > 0
    assert nonempty == B


def test_demod_to_deframer_chain(qrl_ctx):
    """2FSK-1k demodulator ports 2 and 3 -> two type-2 deframers (gr_demod_base.cpp:601-602): the frames of the stream come out
    as 0xB5 + 32 bits, the same bits the oracle deframer extracts from the oracle demodulator's output"""
    import torch
    import qradiolink_amd as q
    iq = sig.make_batch("2fsk1k", 3, nframes=4, device_rate=1000000, seed=21)
    n = iq.shape[1]
    chunk = (n // 3 + 2) & ~1
    dem = q.Demod(qrl_ctx, q.MODEM_2FSK1K, batch=3, max_chunk=chunk)
    dfa, dfb = q.Deframer(qrl_ctx, 2, 3), q.Deframer(qrl_ctx, 2, 3)
    d = torch.from_numpy(iq).cuda()
    got = {k: [[] for _ in range(3)] for k in "ab"}
    for s in range(0, n, chunk):
        part = d[:, s:s + chunk]
        if part.shape[1] & 1:
            part = part[:, :-1]
        dem.process_async(part.contiguous())
        dem.sync()
        cnt = dem.counts.view(torch.int32)
        for key, df, bits, col in (("a", dfa, dem.bits_a, 2), ("b", dfb, dem.bits_b, 3)):
            out, oc = df.process(bits, counts=cnt[:, col:], count_stride=4)
            out, oc = out.cpu().numpy(), oc.cpu().numpy()
            for b in range(3):
                got[key][b].append(out[b, :oc[b]].copy())
    dem.close(); dfa.close(); dfb.close()
    frames = 0
    for b in range(3):
        ref = orc.demod_2fsk(orc.frontend(iq[b, :n & ~1], 1000000, 0.0))   # chunk is even: only the last call can drop a sample
        for key, port in (("a", "bits_a"), ("b", "bits_b")):
            g = np.concatenate(got[key][b])
            w = orc.deframer(2, ref[port])
            assert np.array_equal(g, w), (b, key)
            frames += w.size // 40
    assert frames >= 3 * 3


# ---- gr_modem::synchronize / findSync / packBytes on the device
def _bits_with_frames(rng, modem, nframes, noise_bits=200):
    """random bits with frames of the mode's types planted in them (sync word + payload), like gr_modem::frame builds them"""
    import ctypes
    bl, fl = ctypes.c_int(), ctypes.c_int()
    cls = orc.lib.orc_modem_sync_geometry(modem, ctypes.byref(bl), ctypes.byref(fl))
    words = {0: [(0xB5, 8)], 1: [(0xDE98AA, 24), (0x98DEAA, 24), (0x4C8A2B, 24)],
             2: [(0xED89, 16), (0x89EDAA, 24), (0xED77AA, 24), (0x8CC8DD, 24), (0x4C8A2B, 24)],
             3: [(0x55F7, 16), (0xFF5D, 16), (0x555D555D, 32)]}[cls]
    parts = []
    for _ in range(nframes):
        parts.append(rng.integers(0, 2, int(rng.integers(3, noise_bits)), dtype=np.uint8))
        w, nb = words[int(rng.integers(0, len(words)))]
        parts.append(np.array([(w >> (nb - 1 - k)) & 1 for k in range(nb)], np.uint8))
        parts.append(rng.integers(0, 2, bl.value, dtype=np.uint8))
    parts.append(rng.integers(0, 2, 50, dtype=np.uint8))
    return np.concatenate(parts)


@pytest.mark.parametrize("modem", [18, 22, 17, 26, 27, 5, 40])
@pytest.mark.parametrize("ncuts", [1, 7])
def test_framesync_bit_exact(qrl_ctx, modem, ncuts):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(modem * 10 + ncuts)
    B = 4
    streams = [_bits_with_frames(rng, modem, 6) for _ in range(B)]
    n = min(s.size for s in streams)
    data = np.stack([s[:n] for s in streams])
    cuts = np.sort(rng.choice(np.arange(1, n), size=ncuts - 1, replace=False)) if ncuts > 1 else np.array([], int)
    edges = [0] + [int(c) for c in cuts] + [n]
    fs = q.FrameSync(qrl_ctx, modem, B)
    refs = [orc.ModemSync(modem) for _ in range(B)]
    d = torch.from_numpy(data).cuda()
    got = [[] for _ in range(B)]
    want = [[] for _ in range(B)]
    for a, e in zip(edges[:-1], edges[1:]):
        out, oc = fs.process(d[:, a:e].contiguous())
        out, oc, act = out.cpu().numpy(), oc.cpu().numpy(), fs.activity.cpu().numpy()
        for b in range(B):
            got[b].append(out[b, :oc[b, 0]].copy())
            want[b].append(refs[b].feed_raw(data[b, a:e]))
            # qrl_framesync_set_activity_output: the bits collected while a sync was held in THIS call (gr_modem::synchronize's return value)
            assert int(act[b]) == refs[b].collected, (b, a, e)
    fs.close()
    total = 0
    for b in range(B):
        g, w = np.concatenate(got[b]), np.concatenate(want[b])
        assert g.size == w.size and np.array_equal(g, w), "stream %d" % b
        total += len(orc.parse_frames(w))
    assert total >= B * 3


def test_demod_to_frames_on_device(qrl_ctx):
    """GMSK-10k demodulator port -> device frame synchroniser: the transmitted 47-byte payloads come back as FrameTypeVoice
    records (0xED89 + reserved byte 0xAA + payload), found in one of the two branches"""
    import torch
    import qradiolink_amd as q
    iq = sig.make_batch("gmsk10k", 2, nframes=3, device_rate=1000000, seed=31)
    n = iq.shape[1]
    dem = q.Demod(qrl_ctx, q.MODEM_GMSK10K, batch=2, max_chunk=n)
    fa, fb = q.FrameSync(qrl_ctx, q.MODEM_GMSK10K, 2), q.FrameSync(qrl_ctx, q.MODEM_GMSK10K, 2)
    dem.process_async(torch.from_numpy(iq).cuda())
    dem.sync()
    cnt = dem.counts.view(torch.int32)
    frames = []
    for f, bits, col in ((fa, dem.bits_a, 2), (fb, dem.bits_b, 3)):
        out, oc = f.process(bits, counts=cnt[:, col:], count_stride=4)
        out, oc = out.cpu().numpy(), oc.cpu().numpy()
        frames.append([orc.parse_frames(out[b, :oc[b, 0]]) for b in range(2)])
    dem.close(); fa.close(); fb.close()
    for b in range(2):
        _, payloads = sig.make_stream("gmsk10k", 3, 1000000, 25000.0, 31 + 101 * b, 0.05, lead=37 * b)
        best = max(sum(any(ft == 0xED89 and p[1:] == want for ft, p in frames[k][b]) for want in payloads) for k in (0, 1))
        assert best == len(payloads)
