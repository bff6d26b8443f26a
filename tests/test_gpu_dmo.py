"""a37b on the GPU: qrl_demod_set_dmo_output (k_dmo_sink behind port 3 of the HIP gr_demod_dmr chain) against the oracle
(orc.demod_dmr_port3 -> orc.DmoSink): records bit for bit, over ragged call sizes, plus the known answer: the DMR bursts that
were put on the air come back (all but the last dibit, which the reference reads one lap early -- tests/test_dmo_sink.py)."""
import numpy as np
import pytest

import orc
import sig
import test_dmo_sink as T

pytestmark = pytest.mark.gpu


def _air(seed, kinds, cc):
    rng = np.random.default_rng(seed)
    frames = T._burst(rng, kinds, cc=cc)
    x, _ = sig.make_4fsk(levels=sig.dmr_levels(frames), seed=seed, cfo=20.0 * (seed % 3))
    return frames, x


@pytest.mark.parametrize("chunk", [1 << 19, 100000, 33334])
def test_dmo_sink_bit_exact_and_recovers_the_bursts(qrl_ctx, chunk):
    import torch
    import qradiolink_amd as q
    calls = [["lc", "voice_sync", "v", "v", "v", "v", "v", "voice_sync", "v", "term"], ["csbk", "lc", "voice_sync", "v", "term", "csbk", "csbk", "lc", "v", "v"],
             ["voice_sync", "v", "v", "v", "v", "v", "voice_sync", "v", "v", "term"]]
    sent, xs = zip(*[_air(11 + i, k, 3 + i) for i, k in enumerate(calls)])
    n = min(x.size for x in xs) & ~1
    iq = np.stack([x[:n] for x in xs])
    dem = q.Demod(qrl_ctx, q.MODEM_DMR, batch=3, max_chunk=chunk)
    dem.enable_dmo_sink(cap_frames=32)
    d = torch.from_numpy(iq).cuda()
    got = [[] for _ in range(3)]
    for s in range(0, n, chunk):
        dem.process(d[:, s:s + chunk].contiguous() if (s + chunk <= n) else d[:, s:].contiguous())
        for b, recs in enumerate(dem.dmo_records()):
            got[b].extend(recs)
    dem.close()
    for b in range(3):
        want = orc.DmoSink().process(orc.demod_dmr_port3(iq[b]))
        assert got[b] == want, "stream %d: records differ from the oracle" % b
        assert len(want) >= 7 and all(T._same(g[3], f) for g, f in zip(want, sent[b]))
        if b < 2:   # (stream 2 starts with voice syncs: the colour code is only known from its terminator on)
            assert all(g[2] == 3 + b for g in want)
        else:
            assert [g[2] for g in want] == [0] * 9 + [5]


def test_golay_table_of_the_library_equals_the_oracles(qrl_ctx):
    """the library generates DECODING_TABLE_1987 itself; a data-sync burst with three bit errors inside its slot type must still
    come back with the right colour code on the device (the oracle's table is checked against the reference source on the CPU)"""
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(5)
    f = bytearray(sig.dmr_frame(rng.integers(0, 2, 196), sig.DMR_MS_DATA_SYNC, 9, 0x03))
    for bit in (98, 103, 160):          # three errors in the 20-bit slot type code word
        f[bit >> 3] ^= 0x80 >> (bit & 7)
    x, _ = sig.make_4fsk(levels=sig.dmr_levels([bytes(f)]), seed=2)
    x = x[: x.size & ~1]
    dem = q.Demod(qrl_ctx, q.MODEM_DMR, batch=1, max_chunk=x.size)
    dem.enable_dmo_sink()
    dem.process(torch.from_numpy(x[None, :]).cuda())
    recs = dem.dmo_records()[0]
    dem.close()
    assert recs == orc.DmoSink().process(orc.demod_dmr_port3(x)) and len(recs) == 1 and recs[0][2] == 9
