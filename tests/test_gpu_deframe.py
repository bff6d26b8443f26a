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
