"""GPU parity of the multi-carrier MMDVM receiver (PFB channelizer + per-channel FM chain -> int16) against the
oracle: bit-exact int16, chunk invariance, channel-range sharding (what the ranks of a multi-GPU job do)."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def _wideband(M, n, seed, nstreams):
    """Several FM carriers on the 25 kHz grid + noise, like an MMDVM multi-carrier band."""
    rng = np.random.default_rng(seed)
    fs = 25000.0 * M
    t = np.arange(n)
    out = []
    for s in range(nstreams):
        x = 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        for c in rng.choice(M, size=min(M, 5), replace=False):
            f0 = c * 25000.0 if c <= M // 2 else (c - M) * 25000.0
            dev, fm = rng.uniform(1000, 4000), rng.uniform(200, 1500)
            ph = 2 * np.pi * f0 * t / fs + (dev / fm) * np.sin(2 * np.pi * fm * t / fs + rng.uniform(0, 6))
            x = x + rng.uniform(0.05, 0.3) * np.exp(1j * ph)
        out.append(x.astype(np.complex64))
    return np.stack(out)


def _run(qrl_ctx, iq, M, chunk, c_first=0, c_count=0):
    import torch
    import qradiolink_amd as q
    ch = q.Channelizer(qrl_ctx, M, batch=iq.shape[0], max_chunk=chunk, channel_first=c_first, channel_count=c_count)
    d = torch.from_numpy(iq).cuda()
    parts = []
    for s in range(0, iq.shape[1], chunk):
        out, cnt = ch.process(d[:, s:s + chunk].contiguous())
        cnt = cnt.cpu().numpy()
        o = out.cpu().numpy()
        parts.append([[o[b, c, :cnt[b, c]].copy() for c in range(ch.cc)] for b in range(iq.shape[0])])
    ch.close()
    return [[np.concatenate([p[b][c] for p in parts]) for c in range(len(parts[0][0]))] for b in range(iq.shape[0])]


@pytest.mark.parametrize("M,n", [(10, 10 * 6000), (64, 64 * 2500)])
def test_channelizer_bit_exact(qrl_ctx, M, n):
    iq = _wideband(M, n, seed=M, nstreams=2)
    got = _run(qrl_ctx, iq, M, n)
    for b in range(2):
        ref = orc.demod_mmdvm_multi(iq[b], M)
        for c in range(M):
            assert got[b][c].size == ref.shape[1], (b, c, got[b][c].size, ref.shape)
            assert np.array_equal(got[b][c], ref[c]), "stream %d channel %d differs" % (b, c)
        assert np.abs(ref).max() > 1000   # the FM carriers are actually there


@pytest.mark.parametrize("chunk", [10 * 1000, 10 * 333, 10 * 25])
def test_channelizer_chunk_invariance(qrl_ctx, chunk):
    M, n = 10, 10 * 6000
    iq = _wideband(M, n, seed=3, nstreams=1)
    got = _run(qrl_ctx, iq, M, chunk)
    ref = orc.demod_mmdvm_multi(iq[0], M)
    for c in range(M):
        assert np.array_equal(got[0][c], ref[c])


def test_channel_range_sharding(qrl_ctx):
    """two 'ranks' take channels [0, 32) and [32, 64) of the same wideband input: together they equal the full result"""
    M, n = 64, 64 * 1500
    iq = _wideband(M, n, seed=9, nstreams=1)
    lo = _run(qrl_ctx, iq, M, n, 0, 32)
    hi = _run(qrl_ctx, iq, M, n, 32, 32)
    ref = orc.demod_mmdvm_multi(iq[0], M)
    for c in range(32):
        assert np.array_equal(lo[0][c], ref[c]) and np.array_equal(hi[0][c], ref[32 + c])


def _run_rssi(qrl_ctx, iq, M, chunk, cal=0.0):
    """like _run, also collecting the rssi_tag_block values"""
    import torch
    import qradiolink_amd as q
    ch = q.Channelizer(qrl_ctx, M, batch=iq.shape[0], max_chunk=chunk)
    ch.calibrate_rssi(cal)
    d = torch.from_numpy(iq).cuda()
    B = iq.shape[0]
    samples = [[[] for _ in range(ch.cc)] for _ in range(B)]
    tags = [[[] for _ in range(ch.cc)] for _ in range(B)]
    for s in range(0, iq.shape[1], chunk):
        out, cnt = ch.process(d[:, s:s + chunk].contiguous())
        cnt, o = cnt.cpu().numpy(), out.cpu().numpy()
        rc, r = ch.rssi_counts.cpu().numpy(), ch.rssi.cpu().numpy()
        for b in range(B):
            for c in range(ch.cc):
                samples[b][c].append(o[b, c, :cnt[b, c]].copy())
                tags[b][c].append(r[b, c, :rc[b, c]].copy())
    ch.close()
    cat = lambda L: [[np.concatenate(x) for x in row] for row in L]
    return cat(samples), cat(tags)


@pytest.mark.parametrize("chunk", [10 * 6000, 10 * 777])
def test_channelizer_rssi_tags(qrl_ctx, chunk):
    """rssi_tag_block between filter and discriminator (gr_demod_mmdvm_multi2.cpp:126-127): same 300-sample serial sums as
    the oracle; log10f differs from libm by a few ulp at most -> 1e-4 dB tolerance"""
    M, n = 10, 10 * 6000
    iq = _wideband(M, n, seed=11, nstreams=2)
    got, tags = _run_rssi(qrl_ctx, iq, M, chunk, cal=-12.5)
    for b in range(2):
        ref, rref = orc.demod_mmdvm_multi_rssi(iq[b], M, cal=-12.5)
        for c in range(M):
            assert np.array_equal(got[b][c], ref[c])
            assert tags[b][c].size == rref[c].size == ref.shape[1] // 300
            assert np.allclose(tags[b][c], rref[c], rtol=0, atol=1e-4), (b, c)


@pytest.mark.parametrize("chunk", [125000, 33333, 1250])
def test_single_carrier_mmdvm_chain(qrl_ctx, chunk):
    """gr_demod_mmdvm (num_channels = 1): 250 ksps -> 12/125 resampler -> rssi -> LPF -> discriminator -> int16, bit-exact and
    chunk invariant"""
    rng = np.random.default_rng(5)
    n, fs = 125000, 250000.0
    t = np.arange(n)
    iq = []
    for s in range(2):
        dev, fm = rng.uniform(1000, 4000), rng.uniform(200, 1500)
        ph = (dev / fm) * np.sin(2 * np.pi * fm * t / fs + rng.uniform(0, 6)) + 2 * np.pi * rng.uniform(-500, 500) * t / fs
        iq.append((0.2 * np.exp(1j * ph) + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64))
    iq = np.stack(iq)
    got, tags = _run_rssi(qrl_ctx, iq, 1, chunk, cal=2.0)
    for b in range(2):
        ref, rref = orc.demod_mmdvm(iq[b], cal=2.0)
        assert got[b][0].size == ref.size and np.array_equal(got[b][0], ref)
        assert tags[b][0].size == rref.size and np.allclose(tags[b][0], rref, rtol=0, atol=1e-4)
        assert np.abs(ref).max() > 1000


def _run_4fsk(qrl_ctx, iq, M, chunk):
    import torch
    import qradiolink_amd as q
    ch = q.Channelizer(qrl_ctx, M, batch=iq.shape[0], max_chunk=chunk)
    ch.enable_4fsk()
    d = torch.from_numpy(iq).cuda()
    B = iq.shape[0]
    dib = [[[] for _ in range(ch.cc)] for _ in range(B)]
    for s in range(0, iq.shape[1], chunk):
        ch.process(d[:, s:s + chunk].contiguous())
        cnt, bits = ch.fsk_counts.cpu().numpy(), ch.dibits.cpu().numpy()
        for b in range(B):
            for c in range(ch.cc):
                dib[b][c].append(bits[b, c, :cnt[b, c, 2]].copy())
    ch.close()
    return [[np.concatenate(x) for x in row] for row in dib]


@pytest.mark.parametrize("chunk", [10 * 6000, 10 * 1250])
def test_channelizer_4fsk_tail_bit_exact(qrl_ctx, chunk):
    """BASELINE config 4: channelizer + per-channel 4FSK symbol demodulator (gr_demod_dmr chain behind the channel filter)"""
    import sig
    M, n = 10, 10 * 6000
    fs = 25000.0 * M
    iq = _wideband(M, n, seed=21, nstreams=2)
    # plant true 4FSK carriers (4800 sym/s) on channels 2 and 8 of stream 0
    t = np.arange(n)
    dibs = {}
    for c, seed in ((2, 5), (8, 6)):
        x, d = sig.make_4fsk(nsym=int(n / fs * 4800) - 2, seed=seed, amp=0.4, noise=0.0, fs=fs)
        f0 = c * 25000.0 if c <= M // 2 else (c - M) * 25000.0
        m = min(n, x.size)
        iq[0, :m] += (x[:m] * np.exp(2j * np.pi * f0 * t[:m] / fs)).astype(np.complex64)
        dibs[c] = d
    got = _run_4fsk(qrl_ctx, iq, M, chunk)
    for b in range(2):
        _, ref = orc.demod_mmdvm_multi_4fsk(iq[b], M)
        for c in range(M):
            assert got[b][c].size == ref[c].size and np.array_equal(got[b][c], ref[c]), (b, c)
    # known answer: the planted dibits come back (map of gr_demod_dmr.cpp:81-86 = the DMR air-interface dibits)
    for c, d in dibs.items():
        g = got[0][c].reshape(-1, 2)
        g = g[:, 0] * 2 + g[:, 1]
        best = max(np.mean(g[k:k + 800] == d[:800]) for k in range(60))
        assert best > 0.99, (c, best)


@pytest.mark.parametrize("cuts", [[64 * 2500], [64 * 777, 64 * 1723], [64 * 31, 64 * 1200, 64 * 1269]])
def test_channelizer_64_4fsk_tail_rssi_bit_exact(qrl_ctx, cuts):
    """The configuration bench.py times as C4 (gr_demod_mmdvm_multi2.cpp:58-135 + gr_demod_dmr.cpp:62-105): M = 64, PFB form 0,
    enable_4fsk + RSSI tags -- int16 FM samples and dibits array_equal with the oracle, RSSI to 1e-4 dB, in one call and over ragged
    cuts that are no multiple of the channelizer's 32-instant tile nor of the per-channel 1200-output tile; planted DMR-like 4FSK
    carriers come back as their dibits."""
    import torch
    import qradiolink_amd as q
    import sig
    M, n = 64, sum(cuts)
    fs = 25000.0 * M
    iq = _wideband(M, n, seed=164, nstreams=2)
    t = np.arange(n)
    dibs = {}
    for c, seed in ((3, 5), (33, 6), (62, 7)):     # (channels 54, 7, 57, 44, 41 of stream 0 carry _wideband's FM carriers)
        x, d = sig.make_4fsk(nsym=int(n / fs * 4800) - 2, seed=seed, amp=0.4, noise=0.0, fs=fs)
        f0 = c * 25000.0 if c <= M // 2 else (c - M) * 25000.0
        m = min(n, x.size)
        iq[0, :m] += (x[:m] * np.exp(2j * np.pi * f0 * t[:m] / fs)).astype(np.complex64)
        dibs[c] = d
    ch = q.Channelizer(qrl_ctx, M, batch=2, max_chunk=max(cuts))
    ch.calibrate_rssi(-7.25)
    ch.enable_4fsk()
    d = torch.from_numpy(iq).cuda()
    got = [[[] for _ in range(M)] for _ in range(2)]
    tags = [[[] for _ in range(M)] for _ in range(2)]
    dib = [[[] for _ in range(M)] for _ in range(2)]
    pos = 0
    for cut in cuts:
        out, cnt = ch.process(d[:, pos:pos + cut].contiguous())
        pos += cut
        cnt, o = cnt.cpu().numpy(), out.cpu().numpy()
        rc, r = ch.rssi_counts.cpu().numpy(), ch.rssi.cpu().numpy()
        fc, bits = ch.fsk_counts.cpu().numpy(), ch.dibits.cpu().numpy()
        for b in range(2):
            for c in range(M):
                got[b][c].append(o[b, c, :cnt[b, c]].copy())
                tags[b][c].append(r[b, c, :rc[b, c]].copy())
                dib[b][c].append(bits[b, c, :fc[b, c, 2]].copy())
    ch.close()
    for b in range(2):
        ref, rref, dref = orc.demod_mmdvm_multi_full(iq[b], M, cal=-7.25)
        for c in range(M):
            g = np.concatenate(got[b][c])
            assert g.size == ref.shape[1] and np.array_equal(g, ref[c]), (b, c)
            tg = np.concatenate(tags[b][c])
            assert tg.size == rref[c].size == ref.shape[1] // 300 and np.allclose(tg, rref[c], rtol=0, atol=1e-4), (b, c)
            dd = np.concatenate(dib[b][c])
            assert dd.size == dref[c].size and np.array_equal(dd, dref[c]), (b, c)
    assert np.abs(ref).max() > 1000
    for c, dw in dibs.items():
        g = np.concatenate(dib[0][c]).reshape(-1, 2)
        g = g[:, 0] * 2 + g[:, 1]
        assert max(np.mean(g[k:k + 400] == dw[:400]) for k in range(60)) > 0.99, c


def _wideband_xl(fs, n, seed, nstreams, offsets):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    out = []
    for s in range(nstreams):
        x = 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        for f0 in offsets:
            dev, fm = rng.uniform(1000, 4000), rng.uniform(200, 1500)
            ph = 2 * np.pi * f0 * t / fs + (dev / fm) * np.sin(2 * np.pi * fm * t / fs + rng.uniform(0, 6))
            x = x + rng.uniform(0.05, 0.3) * np.exp(1j * ph)
        out.append(x.astype(np.complex64))
    return np.stack(out)


@pytest.mark.parametrize("N,D,chunk", [(7, 10, 48000), (7, 10, 10002), (16, 32, 76800)])
def test_freq_xlating_form_bit_exact(qrl_ctx, N, D, chunk):
    """form 1 = legacy gr_demod_mmdvm_multi: per channel rotator + 1:D resampler (the front-end MFMA kernel, one launch per
    channel with that channel's exact NCO), LPF, RSSI, discriminator -> int16; bit-exact and chunk invariant"""
    import torch
    import qradiolink_amd as q
    fs = 24000.0 * D
    n = 48000 if D == 10 else 76800 * 2
    iq = _wideband_xl(fs, n, seed=N, nstreams=2, offsets=[0.0, 25000.0, -50000.0, 75000.0])
    ch = q.Channelizer(qrl_ctx, N, batch=2, max_chunk=chunk, form=1, decimation=D)
    ch.calibrate_rssi(1.5)
    d = torch.from_numpy(iq).cuda()
    got = [[[] for _ in range(N)] for _ in range(2)]
    tags = [[[] for _ in range(N)] for _ in range(2)]
    for s in range(0, n, chunk):
        part = d[:, s:s + chunk]
        if part.shape[1] & 1:
            part = part[:, :-1]
        out, cnt = ch.process(part.contiguous())
        cnt, o = cnt.cpu().numpy(), out.cpu().numpy()
        rc, r = ch.rssi_counts.cpu().numpy(), ch.rssi.cpu().numpy()
        for b in range(2):
            for c in range(N):
                got[b][c].append(o[b, c, :cnt[b, c]].copy())
                tags[b][c].append(r[b, c, :rc[b, c]].copy())
    ch.close()
    used = sum((min(chunk, n - s) & ~1) for s in range(0, n, chunk))
    for b in range(2):
        ref, rref = orc.demod_mmdvm_xlating(iq[b, :used], N, D=D, cal=1.5)
        for c in range(N):
            g = np.concatenate(got[b][c])
            assert g.size == ref.shape[1] and np.array_equal(g, ref[c]), (b, c)
            tg = np.concatenate(tags[b][c])
            assert tg.size == rref[c].size and np.allclose(tg, rref[c], rtol=0, atol=1e-4)
    assert np.abs(ref).max() > 1000


# ---- multi-carrier MMDVM transmitter (gr_mod_mmdvm_multi2): FM modulators + 25/24 resamplers + pfb_synthesizer_ccf(10)
def _audio(N, n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    x = np.zeros((N, n), np.int16)
    for c in range(N):
        x[c] = (rng.uniform(3000, 12000) * np.sin(2 * np.pi * rng.uniform(200, 2500) * t / 24000 + rng.uniform(0, 6))
                + rng.normal(0, 300, n)).astype(np.int16)
    return x


@pytest.mark.parametrize("N,cuts", [(7, [24000]), (3, [24000]), (7, [5000, 1, 9999, 9000])])
def test_mmdvm_tx_synthesizer_bit_exact(qrl_ctx, N, cuts):
    import torch
    import qradiolink_amd as q
    n = sum(cuts)
    x = np.stack([_audio(N, n, seed=10 * N + b) for b in range(2)])
    syn = q.Synth(qrl_ctx, N, batch=2, max_samples=max(cuts))
    d = torch.from_numpy(x).cuda()
    parts, pos = [], 0
    for c in cuts:
        parts.append(syn.process(d[:, :, pos:pos + c]).cpu().numpy())
        pos += c
    syn.close()
    got = np.concatenate(parts, axis=1)
    for b in range(2):
        ref = orc.mod_mmdvm_multi(x[b])
        assert got[b].size == ref.size, (got[b].size, ref.size)
        g, w = got[b].view(np.float32) + np.float32(0), ref.view(np.float32) + np.float32(0)
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), "stream %d differs" % b
    assert np.abs(ref).max() > 0.3


def test_mmdvm_tx_rx_loopback_on_gpu(qrl_ctx):
    """HIP synthesizer (7 channels) -> HIP channelizer (10 x 25 kHz): every channel's audio tone comes back on its RX port
    (port map {0,1,2,3,9,8,7} on both sides)"""
    import torch
    import qradiolink_amd as q
    N, n = 7, 48000
    t = np.arange(n)
    x = np.stack([(9000 * np.sin(2 * np.pi * (300 + 150 * c) * t / 24000)).astype(np.int16) for c in range(N)])[None]
    syn = q.Synth(qrl_ctx, N, batch=1, max_samples=n)
    iq = syn.process(torch.from_numpy(x).cuda())
    syn.close()
    m = (iq.shape[1] // 10) * 10
    ch = q.Channelizer(qrl_ctx, 10, batch=1, max_chunk=m)
    out, cnt = ch.process(iq[:, :m].contiguous())
    out, cnt = out.cpu().numpy(), cnt.cpu().numpy()
    ch.close()
    for c in range(N):
        p = c if c <= 3 else 10 - (c - 3)
        r = out[0, p, 4000:4000 + 16384].astype(np.float64)
        f = np.abs(np.fft.rfft(r * np.hanning(r.size)))
        assert abs(np.argmax(f) * 24000 / r.size - (300 + 150 * c)) < 3.0
        assert 0.8 * 9000 < np.percentile(np.abs(r), 99) < 1.1 * 9000


@pytest.mark.parametrize("cuts", [[12000], [1000, 7, 5000, 5993]])
def test_mmdvm_tx_single_carrier_bit_exact(qrl_ctx, cuts):
    """gr_mod_mmdvm (single_carrier): FM -> LPF -> x0.8 -> bb gain -> rational_resampler_ccf(125, 12)"""
    import torch
    import qradiolink_amd as q
    n = sum(cuts)
    x = np.stack([_audio(1, n, seed=50 + b) for b in range(2)])
    syn = q.Synth(qrl_ctx, 1, batch=2, max_samples=max(cuts), bb_gain=0.75, single_carrier=True)
    d = torch.from_numpy(x).cuda()
    parts, pos = [], 0
    for c in cuts:
        parts.append(syn.process(d[:, :, pos:pos + c]).cpu().numpy())
        pos += c
    syn.close()
    got = np.concatenate(parts, axis=1)
    for b in range(2):
        ref = orc.mod_mmdvm(x[b, 0], bb_gain=0.75)
        assert got[b].size == ref.size, (got[b].size, ref.size)
        g, w = got[b].view(np.float32) + np.float32(0), ref.view(np.float32) + np.float32(0)
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    assert np.abs(ref).max() > 0.3


@pytest.mark.parametrize("single", [False, True])
def test_mmdvm_tx_zero_idle_bursts_bit_exact(qrl_ctx, single):
    """gr_zero_idle_bursts (a50): zero_samples runs applied on the device (behind the 25/24 resampler of the multi-carrier graph,
    behind the FM modulator of the single-carrier one) equal the oracle's, over ragged calls and runs that cross call boundaries;
    an idle slot (750 items from a slot boundary) really mutes that channel's carrier."""
    import torch
    import qradiolink_amd as q
    N = 1 if single else 3
    cuts = [720, 1000, 7, 5000, 4273]
    n = sum(cuts)
    x = np.stack([_audio(N, n, seed=70 + b) for b in range(2)])
    runs = [(0, 0, 750, 750), (0, N - 1, 2990, 750), (1, 0, 0, 750), (1, 0, 700, 750), (0, 0, 6000, 100), (0, 0, 6020, 10)]   # (stream, channel, start, count)
    # (the last one starts inside its predecessor and ENDS it at 6030: one down-counter that a tag reloads, gr_zero_idle_bursts.cpp:62-69;
    #  pinned against the reference block in test_ref_blocks.py)
    syn = q.Synth(qrl_ctx, N, batch=2, max_samples=max(cuts), bb_gain=1.0, single_carrier=single)
    syn.add_zero_runs(runs[:3])
    d = torch.from_numpy(x).cuda()
    parts, pos = [], 0
    for i, c in enumerate(cuts):
        if i == 1:
            syn.add_zero_runs(runs[3:])
        parts.append(syn.process(d[:, :, pos:pos + c]).cpu().numpy())
        pos += c
    syn.close()
    got = np.concatenate(parts, axis=1)
    for b in range(2):
        orc.set_zero_runs([(ch, st, cnt) for s, ch, st, cnt in runs if s == b])
        ref = orc.mod_mmdvm(x[b, 0], bb_gain=1.0) if single else orc.mod_mmdvm_multi(x[b])
        orc.set_zero_runs(None)
        assert got[b].size == ref.size
        g, w = got[b].view(np.float32) + np.float32(0), ref.view(np.float32) + np.float32(0)
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), "stream %d differs" % b
    plain = orc.mod_mmdvm(x[1, 0], bb_gain=1.0) if single else orc.mod_mmdvm_multi(x[1])
    assert not np.array_equal(plain, ref)            # the runs did something


# ---- BASELINE.json configs[3] taken literally: 64 x freq-xlating FIR 1:64 (2181-tap prototype) + per-channel chain + 4FSK demod
@pytest.mark.parametrize("chunk", [64 * 2500, 64 * 625 + 2])
def test_literal_c4_freq_xlating_bank_64_channels_bit_exact(qrl_ctx, chunk):
    """form = 2: 64 channels on the 25 kHz grid out of one 1.6 Msps input, each through its own rotator + rational_resampler_ccf(1, 64,
    low_pass_2(1, 1.6e6, 5000, 2000, 60, BH)) (the MFMA decimator, one launch per channel), then 24/25 resampler, LPF, RSSI,
    discriminator -> int16 and the 4FSK symbol tail -- int16, RSSI and dibits bit-exact against the oracle, in one call and cut into
    ragged calls; planted DMR-like 4FSK carriers come back as their dibits; the PFB form (form 0) produces the same dibits from the
    same input (the two forms are the same filter bank, summed in different orders)."""
    import torch
    import qradiolink_amd as q
    import sig
    N, n = 64, 64 * 2500
    fs = 25000.0 * N
    iq = _wideband(N, n, seed=64, nstreams=2)
    t = np.arange(n)
    dibs = {}
    for c, seed in ((3, 5), (40, 6), (63, 7)):
        x, d = sig.make_4fsk(nsym=int(n / fs * 4800) - 2, seed=seed, amp=0.4, noise=0.0, fs=fs)
        f0 = c * 25000.0 if c <= N // 2 else (c - N) * 25000.0
        m = min(n, x.size)
        iq[0, :m] += (x[:m] * np.exp(2j * np.pi * f0 * t[:m] / fs)).astype(np.complex64)
        dibs[c] = d
    ch = q.Channelizer(qrl_ctx, N, batch=2, max_chunk=chunk, form=2)
    ch.calibrate_rssi(-3.0)
    ch.enable_4fsk()
    d = torch.from_numpy(iq).cuda()
    got = [[[] for _ in range(N)] for _ in range(2)]
    tags = [[[] for _ in range(N)] for _ in range(2)]
    dib = [[[] for _ in range(N)] for _ in range(2)]
    for s in range(0, n, chunk):
        part = d[:, s:s + chunk]
        if part.shape[1] & 1:
            part = part[:, :-1]
        out, cnt = ch.process(part.contiguous())
        cnt, o = cnt.cpu().numpy(), out.cpu().numpy()
        rc, r = ch.rssi_counts.cpu().numpy(), ch.rssi.cpu().numpy()
        fc, bits = ch.fsk_counts.cpu().numpy(), ch.dibits.cpu().numpy()
        for b in range(2):
            for c in range(N):
                got[b][c].append(o[b, c, :cnt[b, c]].copy())
                tags[b][c].append(r[b, c, :rc[b, c]].copy())
                dib[b][c].append(bits[b, c, :fc[b, c, 2]].copy())
    ch.close()
    used = sum((min(chunk, n - s) & ~1) for s in range(0, n, chunk))
    for b in range(2):
        ref, rref, dref = orc.demod_mmdvm_xlating_bank_4fsk(iq[b, :used], N, cal=-3.0)
        for c in range(N):
            g = np.concatenate(got[b][c])
            assert g.size == ref.shape[1] and np.array_equal(g, ref[c]), (b, c)
            tg = np.concatenate(tags[b][c])
            assert tg.size == rref[c].size and np.allclose(tg, rref[c], rtol=0, atol=1e-4)
            dd = np.concatenate(dib[b][c])
            assert dd.size == dref[c].size and np.array_equal(dd, dref[c]), (b, c)
    assert np.abs(ref).max() > 1000
    for c, dw in dibs.items():
        g = np.concatenate(dib[0][c]).reshape(-1, 2)
        g = g[:, 0] * 2 + g[:, 1]
        best = max(np.mean(g[k:k + 400] == dw[:400]) for k in range(60))
        assert best > 0.99, (c, best)
    # the PFB form on the same input: same channels, same planted dibits
    if chunk == n:
        pfb = _run_4fsk(qrl_ctx, iq, N, n)
        for c, dw in dibs.items():
            g = pfb[0][c].reshape(-1, 2)
            g = g[:, 0] * 2 + g[:, 1]
            assert max(np.mean(g[k:k + 400] == dw[:400]) for k in range(60)) > 0.99


@pytest.mark.parametrize("chunk", [64 * 2500, 64 * 777])
def test_channelizer_64_legacy_kernel_and_ragged_calls(qrl_ctx, chunk):
    """M = 64 runs on the streaming kernel k_pfb_stream64 by default; QRL_CHAN_OPT_LEGACY_PFB = 1 keeps the general-M kernel and = 2
    the tiled k_pfb_chan64 of round 3 reachable for A/B runs.  All three must equal the oracle, also when the stream is cut into calls
    that are not a multiple of the 16- / 32-instant tiles (history path + ragged last tile)."""
    import torch
    import qradiolink_amd as q
    M, n = 64, 64 * 2500
    iq = _wideband(M, n, seed=77, nstreams=2)
    ref = [orc.demod_mmdvm_multi(iq[b], M) for b in range(2)]
    for legacy in (0, 1):
        ch = q.Channelizer(qrl_ctx, M, batch=2, max_chunk=chunk)
        ch.set_option(q.CHAN_OPT_LEGACY_PFB, legacy)
        d = torch.from_numpy(iq).cuda()
        parts = []
        for s in range(0, n, chunk):
            out, cnt = ch.process(d[:, s:s + chunk].contiguous())
            cnt, o = cnt.cpu().numpy(), out.cpu().numpy()
            parts.append([[o[b, c, :cnt[b, c]].copy() for c in range(M)] for b in range(2)])
        ch.close()
        for b in range(2):
            for c in range(M):
                g = np.concatenate([p[b][c] for p in parts])
                assert g.size == ref[b].shape[1] and np.array_equal(g, ref[b][c]), (legacy, b, c)


@pytest.mark.parametrize("chunk", [10 * 6000, 10 * 1111])
def test_legacy_per_channel_kernels_still_bit_exact(qrl_ctx, chunk):
    """QRL_CHAN_OPT_LEGACY_TAIL = 1: the per-channel chain as the separate kernels of round 2 (k_resamp, k_fir_ccf, k_rssi_tag,
    k_quad_demod, k_fir_fff) instead of the fused k_chan_tail -- int16, RSSI tags and dibits against the oracle"""
    import torch
    import qradiolink_amd as q
    M, n = 10, 10 * 6000
    iq = _wideband(M, n, seed=31, nstreams=2)
    ch = q.Channelizer(qrl_ctx, M, batch=2, max_chunk=chunk)
    ch.set_option(q.CHAN_OPT_LEGACY_TAIL, 1)
    ch.calibrate_rssi(0.5)
    ch.enable_4fsk()
    d = torch.from_numpy(iq).cuda()
    got = [[[] for _ in range(M)] for _ in range(2)]
    tags = [[[] for _ in range(M)] for _ in range(2)]
    dib = [[[] for _ in range(M)] for _ in range(2)]
    for s in range(0, n, chunk):
        out, cnt = ch.process(d[:, s:s + chunk].contiguous())
        cnt, o = cnt.cpu().numpy(), out.cpu().numpy()
        rc, r = ch.rssi_counts.cpu().numpy(), ch.rssi.cpu().numpy()
        fc, bits = ch.fsk_counts.cpu().numpy(), ch.dibits.cpu().numpy()
        for b in range(2):
            for c in range(M):
                got[b][c].append(o[b, c, :cnt[b, c]].copy())
                tags[b][c].append(r[b, c, :rc[b, c]].copy())
                dib[b][c].append(bits[b, c, :fc[b, c, 2]].copy())
    ch.close()
    for b in range(2):
        ref, rref = orc.demod_mmdvm_multi_rssi(iq[b], M, cal=0.5)
        _, dref = orc.demod_mmdvm_multi_4fsk(iq[b], M)
        for c in range(M):
            assert np.array_equal(np.concatenate(got[b][c]), ref[c]), (b, c)
            assert np.allclose(np.concatenate(tags[b][c]), rref[c], rtol=0, atol=1e-4)
            assert np.array_equal(np.concatenate(dib[b][c]), dref[c]), (b, c)


@pytest.mark.parametrize("serial", [0, 1])
def test_channelizer_64_calls_queued_back_to_back_bit_exact(qrl_ctx, serial):
    """Round 5: the per-channel kernel of call k runs on an internal stream beside the channelizer of call k + 1 (the channel ring
    holds two calls).  Five calls are QUEUED without a synchronisation in between, every call with its own output buffers (through the
    C ABI directly: the pointers are kernel arguments of the call that was given them), one qrl_chan_sync at the end -- int16, dibits
    and RSSI tags of the concatenation must equal the oracle's; QRL_CHAN_OPT_SERIAL_TAIL = 1 (the order of rounds 1-4) as well."""
    import ctypes as C
    import torch
    import qradiolink_amd as q
    M = 64
    cuts = [64 * 1500, 64 * 1500, 64 * 777, 64 * 1500, 64 * 1223]
    n = sum(cuts)
    iq = _wideband(M, n, seed=501, nstreams=2)
    ch = q.Channelizer(qrl_ctx, M, batch=2, max_chunk=max(cuts))
    ch.calibrate_rssi(1.5)
    ch.enable_4fsk()
    ch.set_option(q.CHAN_OPT_SERIAL_TAIL, serial)
    assert len(ch.internal_streams) == 3
    d = torch.from_numpy(iq).cuda()
    lib = ch.lib
    bufs = []
    pos = 0
    torch.cuda.synchronize()
    for cut in cuts:
        o = torch.zeros_like(ch.out); cn = torch.zeros_like(ch.counts)
        r = torch.zeros_like(ch.rssi); rc = torch.zeros_like(ch.rssi_counts)
        db = torch.zeros_like(ch.dibits); fc = torch.zeros_like(ch.fsk_counts)
        torch.cuda.synchronize()
        bufs.append((o, cn, r, rc, db, fc))
    for cut, (o, cn, r, rc, db, fc) in zip(cuts, bufs):
        assert lib.qrl_chan_set_rssi_output(ch.h, C.c_void_p(r.data_ptr()), ch.rssi_cap, C.c_void_p(rc.data_ptr())) == 0
        assert lib.qrl_chan_set_4fsk_output(ch.h, C.c_void_p(db.data_ptr()), ch.fsk_cap, None, 0, C.c_void_p(fc.data_ptr())) == 0
        x = d[:, pos:pos + cut]
        assert lib.qrl_chan_process(ch.h, C.c_void_p(x.data_ptr()), x.stride(0), cut, C.c_void_p(o.data_ptr()), ch.cap, C.c_void_p(cn.data_ptr())) == 0
        pos += cut
    ch.sync()
    got = [[[] for _ in range(M)] for _ in range(2)]
    tags = [[[] for _ in range(M)] for _ in range(2)]
    dib = [[[] for _ in range(M)] for _ in range(2)]
    for (o, cn, r, rc, db, fc) in bufs:
        o, cn, r, rc, db, fc = (t.cpu().numpy() for t in (o, cn, r, rc, db, fc))
        for b in range(2):
            for c in range(M):
                got[b][c].append(o[b, c, :cn[b, c]].copy())
                tags[b][c].append(r[b, c, :rc[b, c]].copy())
                dib[b][c].append(db[b, c, :fc[b, c, 2]].copy())
    ch.close()
    for b in range(2):
        ref, rref, dref = orc.demod_mmdvm_multi_full(iq[b], M, cal=1.5)
        for c in range(M):
            g = np.concatenate(got[b][c])
            assert g.size == ref.shape[1] and np.array_equal(g, ref[c]), (serial, b, c)
            tg = np.concatenate(tags[b][c])
            assert tg.size == rref[c].size and np.allclose(tg, rref[c], rtol=0, atol=1e-4), (serial, b, c)
            assert np.array_equal(np.concatenate(dib[b][c]), dref[c]), (serial, b, c)


@pytest.mark.parametrize("ppm,cuts", [(20.0, [64 * 5000]), (20.0, [64 * 1200 + 64, 64 * 2200, 64 * 1598 + 2 * 64 - 64]), (-14000.0, [64 * 2000, 64 * 3000]), (14000.0, [64 * 5000])])
def test_channelizer_64_4fsk_channels_with_symbol_clock_error(qrl_ctx, ppm, cuts):
    """SURVEY 8(d)'s timing impairments on the multi-carrier receiver (VERDICT r5 "missing" #3 for C4's symbol tail): DMR-like 4FSK on five channels whose
    transmitters' symbol clocks are off by `ppm` and whose first symbols sit a fraction of a symbol late -- symbol_sync_ff(TED_MUELLER_AND_MULLER, 5, 2 pi / 100,
    1.0, 0.2869, 0.06, ...) of gr_demod_dmr.cpp:70-71 slides (20 ppm) or sits in its +- 0.06 limiter (1.2 % of 5 samples: -14000 / +14000 ppm; the limiter hits
    are counted on the CPU in tests/test_channel_8d.py).  int16, RSSI and dibits equal the oracle's, in one call and cut into calls."""
    import torch
    import qradiolink_amd as q
    import sig
    M, n = 64, sum(cuts)
    fs = 25000.0 * M
    iq = _wideband(M, n, seed=265, nstreams=2)
    t = np.arange(n)
    for b in range(2):
        for k, c in enumerate((2, 17, 31, 45, 60)):
            x, _ = sig.make_4fsk(nsym=int(n / fs * 4800) - 8, seed=50 + 10 * b + k, amp=0.3, noise=0.0, fs=fs, clock_ppm=ppm * (1 if k % 2 == 0 else -1), frac_delay=0.37 + 0.11 * k)
            f0 = c * 25000.0 if c <= M // 2 else (c - M) * 25000.0
            m = min(n, x.size)
            iq[b, :m] += (x[:m] * np.exp(2j * np.pi * f0 * t[:m] / fs)).astype(np.complex64)
    ch = q.Channelizer(qrl_ctx, M, batch=2, max_chunk=max(cuts))
    ch.calibrate_rssi(0.5)
    ch.enable_4fsk()
    d = torch.from_numpy(iq).cuda()
    got = [[[] for _ in range(M)] for _ in range(2)]
    dib = [[[] for _ in range(M)] for _ in range(2)]
    pos = 0
    for cut in cuts:
        out, cnt = ch.process(d[:, pos:pos + cut].contiguous())
        pos += cut
        cnt, o = cnt.cpu().numpy(), out.cpu().numpy()
        fc, bits = ch.fsk_counts.cpu().numpy(), ch.dibits.cpu().numpy()
        for b in range(2):
            for c in range(M):
                got[b][c].append(o[b, c, :cnt[b, c]].copy())
                dib[b][c].append(bits[b, c, :fc[b, c, 2]].copy())
    ch.close()
    for b in range(2):
        ref, _, dref = orc.demod_mmdvm_multi_full(iq[b], M, cal=0.5)
        for c in range(M):
            g = np.concatenate(got[b][c])
            assert g.size == ref.shape[1] and np.array_equal(g, ref[c]), (b, c)
            dd = np.concatenate(dib[b][c])
            assert dd.size == dref[c].size and np.array_equal(dd, dref[c]), (b, c)
