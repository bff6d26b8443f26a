"""GPU parity of the TX path (gr_mod_qpsk on HIP) against the oracle, TX->RX loopback on the GPU, and
full duplex: modulator and demodulator running concurrently on separate HIP streams (BASELINE config 5)."""
import numpy as np
import pytest

import orc
import sig

pytestmark = pytest.mark.gpu


def _payload(rng, nbytes):
    return rng.integers(0, 256, nbytes, dtype=np.uint8)


@pytest.mark.parametrize("nbytes", [1, 7, 64, 1516 + 11, 8192])
def test_mod_qpsk_bit_exact(qrl_ctx, nbytes):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(nbytes)
    data = np.stack([_payload(rng, nbytes) for _ in range(3)])
    mod = q.Mod(qrl_ctx, q.MODEM_QPSK250K, batch=3, max_bytes=nbytes)
    out = mod.process(torch.from_numpy(data).cuda()).cpu().numpy()
    mod.close()
    for b in range(3):
        ref = orc.mod_qpsk(data[b])
        assert out[b].size == ref.size == nbytes * 32
        assert np.array_equal(out[b].view(np.uint32), ref.view(np.uint32)), "stream %d differs" % b


@pytest.mark.parametrize("cuts", [[5, 1, 300, 2000], [1024, 1024, 1024], [3, 3, 3, 3, 3, 3]])
def test_mod_chunk_invariance(qrl_ctx, cuts):
    """scrambler register, encoder history, differential symbol and pulse-shaping history carry across calls"""
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(5)
    total = sum(cuts)
    data = np.stack([_payload(rng, total) for _ in range(2)])
    mod = q.Mod(qrl_ctx, q.MODEM_QPSK250K, batch=2, max_bytes=max(cuts))
    parts, pos = [], 0
    d = torch.from_numpy(data).cuda()
    for c in cuts:
        parts.append(mod.process(d[:, pos:pos + c].contiguous()).cpu().numpy())
        pos += c
    mod.close()
    got = np.concatenate(parts, axis=1)
    for b in range(2):
        ref = orc.mod_qpsk(data[b])
        assert np.array_equal(got[b].view(np.uint32), ref.view(np.uint32))


def _frames(nframes, rng):
    data, payloads = sig.frames("qpsk250k", nframes, rng)
    return data, payloads


def test_tx_rx_loopback_on_gpu(qrl_ctx):
    """HIP modulator -> (scale) -> HIP demodulator returns the transmitted 1516-byte frames."""
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(9)
    data, payloads = _frames(3, rng)
    data = np.concatenate([data, np.full(64, 0xAA, np.uint8)])
    mod = q.Mod(qrl_ctx, q.MODEM_QPSK250K, batch=1, max_bytes=data.size)
    iq = mod.process(torch.from_numpy(data[None, :]).cuda())
    mod.close()
    iq = (iq * 0.3).contiguous()
    n = iq.shape[1] & ~1
    dem = q.Demod(qrl_ctx, q.MODEM_QPSK250K, batch=1, max_chunk=n)
    out = q.collect(dem, iq[:, :n], n)
    dem.close()
    fr = sig.find_frames(out["bits_a"][0], bytes([0xDE, 0x98, 0xAA]), 1516 * 8)
    assert sum(p in fr for p in payloads) == len(payloads)


def test_full_duplex_two_streams(qrl_ctx):
    """TX and RX handles own different HIP streams; interleaved un-synchronised calls give the same results as
    running each alone (RX and TX top blocks are independent)."""
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(11)
    B = 16
    tx_data = np.stack([_payload(rng, 4096) for _ in range(B)])
    rx_iq = sig.make_batch("qpsk250k", B, nframes=2, device_rate=1000000, seed=40)
    n = rx_iq.shape[1]
    d_tx, d_rx = torch.from_numpy(tx_data).cuda(), torch.from_numpy(rx_iq).cuda()
    mod = q.Mod(qrl_ctx, q.MODEM_QPSK250K, batch=B, max_bytes=1024)
    dem = q.Demod(qrl_ctx, q.MODEM_QPSK250K, batch=B, max_chunk=n // 4 + 2)
    assert mod.lib.qrl_mod_stream(mod.h) != dem.lib.qrl_demod_stream(dem.h)
    tx_parts, rx_bits = [], [[] for _ in range(B)]
    step = (n // 4) & ~1
    for k in range(4):
        tx_parts.append(mod.process_async(d_tx[:, 1024 * k:1024 * (k + 1)].contiguous()))
        part = d_rx[:, step * k:step * (k + 1)].contiguous()
        dem.process_async(part)                 # no sync between the TX and the RX call
        dem.sync()
        cnt = dem.counts.cpu().numpy()
        bits = dem.bits_a.cpu().numpy()
        for b in range(B):
            rx_bits[b].append(bits[b, :cnt[b, 2]].copy())
    mod.sync()
    tx = torch.cat(tx_parts, dim=1).cpu().numpy()
    mod.close()
    dem.close()
    for b in range(0, B, 5):
        assert np.array_equal(tx[b].view(np.uint32), orc.mod_qpsk(tx_data[b]).view(np.uint32))
        ref = orc.demod_qpsk(orc.frontend(rx_iq[b, :4 * step], 1000000, 0.0))
        assert np.array_equal(np.concatenate(rx_bits[b]), ref["bits_a"])


# ---- FSK family modulators (gr_mod_2fsk incl. FM variants, gr_mod_gmsk)
FSK_CASES = [
    # modem, oracle call, bytes
    (18, lambda d: orc.mod_2fsk(d, sps=50, filter_width=2000, fm=False), 24),    # 2FSK1K   (gr_mod_base.cpp:158)
    (16, lambda d: orc.mod_2fsk(d, sps=50, filter_width=2500, fm=True), 24),     # 2FSK1KFM (:156)
    (17, lambda d: orc.mod_2fsk(d, sps=25, filter_width=4000, fm=False), 40),    # 2FSK2K
    (19, lambda d: orc.mod_2fsk(d, sps=5, filter_width=25000, fm=True), 200),    # 2FSK10KFM
    (22, lambda d: orc.mod_gmsk(d, sps=10, filter_width=20000), 300),            # GMSK10K  (:162)
    (20, lambda d: orc.mod_gmsk(d, sps=50, filter_width=4000), 40),              # GMSK2K
    (21, lambda d: orc.mod_gmsk(d, sps=100, filter_width=2000), 20),             # GMSK1K
    (5, lambda d: orc.mod_4fsk(d, sps=25, filter_width=3500, fm=True), 40),      # 4FSK2KFM (gr_mod_base.cpp:164)
    (6, lambda d: orc.mod_4fsk(d, sps=50, filter_width=2000, fm=True), 20),      # 4FSK1KFM
    (4, lambda d: orc.mod_4fsk(d, sps=5, filter_width=20000, fm=True), 200),     # 4FSK10KFM
    (27, lambda d: orc.mod_4fsk(d, sps=2, filter_width=125000, fm=True), 2000),  # 4FSK100K (sps 2 -> 5 x 2)
    (24, lambda d: orc.mod_bpsk(d, sps=500, filter_width=1500), 24),             # BPSK1K (:168): 5501-tap RRC interpolator
    (0, lambda d: orc.mod_bpsk(d, sps=250, filter_width=2800), 40),              # BPSK2K
    (7, lambda d: orc.mod_qpsk(d, sps=500, filter_width=1300), 40),              # QPSK2K (gr_mod_base.cpp:173): 11 x 500 tap RRC
    (1, lambda d: orc.mod_qpsk(d, sps=100, filter_width=6500), 100),             # QPSK20K (:174): 13 x 100 taps
    (3, lambda d: orc.mod_4fsk(d, sps=25, filter_width=4000, fm=False), 40),     # 4FSK2K (:163): repeat(sps), spacing 2
]


@pytest.mark.parametrize("modem,oracle,nbytes", FSK_CASES, ids=[str(c[0]) for c in FSK_CASES])
def test_mod_fsk_family_bit_exact(qrl_ctx, modem, oracle, nbytes):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(modem)
    data = np.stack([_payload(rng, nbytes) for _ in range(3)])
    mod = q.Mod(qrl_ctx, modem, batch=3, max_bytes=nbytes)
    out = mod.process(torch.from_numpy(data).cuda()).cpu().numpy()
    mod.close()
    for b in range(3):
        ref = oracle(data[b])
        assert out[b].size == ref.size, (out[b].size, ref.size)
        assert np.array_equal(out[b].view(np.uint32), ref.view(np.uint32)), "stream %d differs" % b


@pytest.mark.parametrize("modem,oracle", [(18, FSK_CASES[0][1]), (22, FSK_CASES[4][1]), (5, FSK_CASES[7][1]), (24, FSK_CASES[11][1])],
                         ids=["2fsk1k", "gmsk10k", "4fsk2kfm", "bpsk1k"])
def test_mod_fsk_chunk_invariance(qrl_ctx, modem, oracle):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(77)
    cuts = [3, 1, 17, 9]
    data = _payload(rng, sum(cuts))[None, :]
    mod = q.Mod(qrl_ctx, modem, batch=1, max_bytes=max(cuts))
    d = torch.from_numpy(data).cuda()
    parts, pos = [], 0
    for c in cuts:
        parts.append(mod.process(d[:, pos:pos + c].contiguous()).cpu().numpy())
        pos += c
    mod.close()
    got = np.concatenate(parts, axis=1)[0]
    assert np.array_equal(got.view(np.uint32), oracle(data[0]).view(np.uint32))


@pytest.mark.parametrize("mode,modem,sync,nbits", [("gmsk10k", 22, bytes([0xED, 0x89]), 384), ("2fsk1k", 18, bytes([0xB5]), 32)])
def test_fsk_tx_rx_loopback_on_gpu(qrl_ctx, mode, modem, sync, nbits):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(13)
    data, payloads = sig.frames(mode, 3, rng)
    mod = q.Mod(qrl_ctx, modem, batch=1, max_bytes=data.size)
    iq = mod.process(torch.from_numpy(data[None, :]).cuda())
    mod.close()
    iq = (iq * 0.2).contiguous()
    n = iq.shape[1] & ~1
    dem = q.Demod(qrl_ctx, modem, batch=1, max_chunk=n)
    out = q.collect(dem, iq[:, :n], n)
    dem.close()
    best = 0
    for k in ("bits_a", "bits_b"):
        fr = sig.find_frames(out[k][0], sync, nbits)
        best = max(best, sum(((bytes([0xAA]) + p) if mode == "gmsk10k" else p) in fr for p in payloads))
    assert best >= len(payloads) - 1   # the last frame may sit in the decoder's look-ahead


# ---- gr_mod_base back end: rotator at 1 Msps + interpolation to the device rate (gr_mod_base.cpp:38,215-258)
def _back_end_ref(x1, device_rate, offset_hz):
    inc = orc.phase_inc_to_turn(2 * np.pi * offset_hz / 1000000.0)
    return orc.tx_interp(orc.rotator(x1, inc), device_rate)


@pytest.mark.parametrize("modem,oracle,nbytes,rate,offset", [
    (26, lambda d: orc.mod_qpsk(d), 600, 4000000, 25000.0),
    (26, lambda d: orc.mod_qpsk(d), 300, 1000000, -12500.0),       # rotator only
    (22, FSK_CASES[4][1], 60, 10000000, 50000.0),                  # 2090 taps: taps read from global memory
    (18, FSK_CASES[0][1], 4, 2000000, 0.0),                        # resampler only
], ids=["qpsk-4M", "qpsk-1M-rot", "gmsk10k-10M", "2fsk1k-2M"])
def test_mod_back_end_bit_exact(qrl_ctx, modem, oracle, nbytes, rate, offset):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(nbytes)
    data = np.stack([_payload(rng, nbytes) for _ in range(2)])
    mod = q.Mod(qrl_ctx, modem, batch=2, max_bytes=nbytes, device_samp_rate=rate, carrier_offset_hz=offset)
    out = mod.process(torch.from_numpy(data).cuda()).cpu().numpy()
    mod.close()
    for b in range(2):
        ref = _back_end_ref(oracle(data[b]), rate, offset)
        assert out[b].size == ref.size, (out[b].size, ref.size)
        assert np.array_equal(out[b].view(np.uint32), ref.view(np.uint32)), "stream %d differs" % b


def test_mod_back_end_chunks_and_retune(qrl_ctx):
    """history of the back-end interpolator and the NCO phase carry across calls; set_carrier_offset is
    phase-continuous like rotator_cc::set_phase_inc (gr_mod_base.cpp:799-805)"""
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(77)
    cuts = [100, 1, 333, 66]
    data = _payload(rng, sum(cuts))[None, :]
    mod = q.Mod(qrl_ctx, q.MODEM_QPSK250K, batch=1, max_bytes=max(cuts), device_samp_rate=5000000, carrier_offset_hz=10000.0)
    d = torch.from_numpy(data).cuda()
    parts, pos = [], 0
    for i, c in enumerate(cuts):
        if i == 2:
            mod.set_carrier_offset(-30000.0)
        parts.append(mod.process(d[:, pos:pos + c].contiguous()).cpu().numpy())
        pos += c
    mod.close()
    got = np.concatenate(parts, axis=1)[0]
    x1 = orc.mod_qpsk(data[0])
    k = (cuts[0] + cuts[1]) * 32
    inc0 = orc.phase_inc_to_turn(2 * np.pi * 10000.0 / 1e6)
    inc1 = orc.phase_inc_to_turn(2 * np.pi * -30000.0 / 1e6)
    rot = np.concatenate([orc.rotator(x1[:k], inc0), orc.rotator(x1[k:], inc1, (k * inc0) & (2 ** 64 - 1))])
    ref = orc.tx_interp(rot, 5000000)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_device_rate_tx_rx_loopback(qrl_ctx):
    """TX at 4 Msps with +25 kHz offset -> RX at 4 Msps tuned to the same offset: frames come back (the first one is spent on
    acquiring the constant phase the two filter delays leave behind -- the oracle chain loses it too)."""
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(10)
    data, payloads = _frames(3, rng)
    data = np.concatenate([data, np.full(400, 0xAA, np.uint8)])   # flush back-end + front-end filters and the Viterbi frames
    mod = q.Mod(qrl_ctx, q.MODEM_QPSK250K, batch=1, max_bytes=data.size, device_samp_rate=4000000, carrier_offset_hz=25000.0)
    iq = mod.process(torch.from_numpy(data[None, :]).cuda())
    mod.close()
    iq = (iq * 0.3).contiguous()
    n = iq.shape[1] & ~7
    dem = q.Demod(qrl_ctx, q.MODEM_QPSK250K, batch=1, max_chunk=n, device_samp_rate=4000000, carrier_offset_hz=25000.0)
    out = q.collect(dem, iq[:, :n], n)
    dem.close()
    fr = sig.find_frames(out["bits_a"][0], bytes([0xDE, 0x98, 0xAA]), 1516 * 8)
    assert payloads[1] in fr and payloads[2] in fr


# ---- analogue voice modulator (gr_mod_nbfm): audio in, 125 IQ samples per audio sample out
@pytest.mark.parametrize("modem,fw", [(9, 5000), (8, 2500)])
@pytest.mark.parametrize("chunk", [8000, 1000, 324])
def test_nbfm_modulator_bit_exact(qrl_ctx, modem, fw, chunk):
    import torch
    import qradiolink_amd as q
    n = 8000
    t = np.arange(n) / 8000.0
    audio = np.stack([0.5 * np.sin(2 * np.pi * 700 * t) + 0.2 * np.sin(2 * np.pi * 1900 * t),
                      np.random.default_rng(3).uniform(-0.7, 0.7, n)]).astype(np.float32)
    mod = q.AMod(qrl_ctx, modem, batch=2, max_samples=chunk, bb_gain=0.75)
    parts = [mod.process(torch.from_numpy(np.ascontiguousarray(audio[:, s:s + chunk])).cuda()).cpu().numpy() for s in range(0, n, chunk)]
    mod.close()
    got = np.concatenate(parts, axis=1)
    assert got.shape == (2, 125 * n)
    for b in range(2):
        want = orc.mod_nbfm(audio[b], filter_width=fw, bb_gain=0.75)
        g, w = got[b].view(np.float32) + np.float32(0), want.view(np.float32) + np.float32(0)
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), "stream %d differs" % b


def test_nbfm_voice_loopback_on_gpu(qrl_ctx):
    """gr_mod_nbfm -> (level) -> gr_demod_nbfm, both on the device: the tone that went in comes out"""
    import torch
    import qradiolink_amd as q
    n = 8000
    audio = (0.5 * np.sin(2 * np.pi * 700 * np.arange(n) / 8000.0)).astype(np.float32)
    mod = q.AMod(qrl_ctx, q.MODEM_NBFM5000, batch=1, max_samples=n)
    iq = mod.process(torch.from_numpy(audio[None, :]).cuda()) * 0.05
    mod.close()
    dem = q.Demod(qrl_ctx, q.MODEM_NBFM5000, batch=1, max_chunk=iq.shape[1])
    out = q.collect(dem, iq.contiguous(), iq.shape[1])
    dem.close()
    a = out["audio"][0][2000:7000].astype(np.float64)
    spec = np.abs(np.fft.rfft(a * np.hanning(a.size)))
    assert abs(np.argmax(spec) * 8000.0 / a.size - 700.0) < 3.0 and np.sqrt(np.mean(a ** 2)) > 0.3
    with pytest.raises(q.QrlError):
        q.AMod(qrl_ctx, q.MODEM_WBFM, batch=1, max_samples=64)            # no WBFM transmitter in the reference's mode table either


# ---- gr_mod_am (src/gr/gr_mod_am.cpp:26-74): agc2_ff -> rail -> band-pass -> + carrier -> 1:125 -> gains -> 4545-tap complex band-pass
@pytest.mark.parametrize("chunk", [800, 250, 37])
def test_am_modulator_bit_exact(qrl_ctx, chunk):
    import torch
    import qradiolink_amd as q
    n = 800
    t = np.arange(n) / 8000.0
    audio = np.stack([0.5 * np.sin(2 * np.pi * 700 * t) + 0.2 * np.sin(2 * np.pi * 1900 * t),
                      np.random.default_rng(4).uniform(-1.4, 1.4, n)]).astype(np.float32)      # (the second stream drives the rail and the AGC)
    mod = q.AMod(qrl_ctx, q.MODEM_AM5000, batch=2, max_samples=chunk, bb_gain=0.75)
    parts = [mod.process(torch.from_numpy(np.ascontiguousarray(audio[:, s:s + chunk])).cuda()).cpu().numpy()[:, :125 * min(chunk, n - s)] for s in range(0, n, chunk)]
    mod.close()
    got = np.concatenate(parts, axis=1)
    assert got.shape == (2, 125 * n)
    for b in range(2):
        want = orc.mod_am(audio[b], bb_gain=0.75)
        g, w = got[b].view(np.float32) + np.float32(0), want.view(np.float32) + np.float32(0)
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), "stream %d differs" % b
    assert 0.15 < np.abs(got[0, 60000:]).mean() < 0.22          # carrier 0.5 x 0.5 x 0.75


def test_am_voice_loopback_on_gpu(qrl_ctx):
    """gr_mod_am -> (level) -> gr_demod_am, both on the device: the tone that went in comes out"""
    import torch
    import qradiolink_amd as q
    n = 4000
    audio = (0.5 * np.sin(2 * np.pi * 700 * np.arange(n) / 8000.0)).astype(np.float32)
    mod = q.AMod(qrl_ctx, q.MODEM_AM5000, batch=1, max_samples=n)
    iq = mod.process(torch.from_numpy(audio[None, :]).cuda()) * 0.2
    mod.close()
    dem = q.Demod(qrl_ctx, q.MODEM_AM5000, batch=1, max_chunk=iq.shape[1])
    out = q.collect(dem, iq.contiguous(), iq.shape[1])
    dem.close()
    a = out["audio"][0][1500:3500].astype(np.float64)
    spec = np.abs(np.fft.rfft((a - a.mean()) * np.hanning(a.size)))
    assert abs(np.argmax(spec) * 8000.0 / a.size - 700.0) < 6.0 and np.std(a) > 0.02


@pytest.mark.parametrize("modem,sb", [(11, 0), (12, 1)])
@pytest.mark.parametrize("chunk", [1 << 14, 1024, 1000, 333])
def test_ssb_modulator_bit_exact(qrl_ctx, modem, sb, chunk):
    """gr_mod_ssb: audio band-pass, cessb clipper + stretcher (whole 1024-chunks, two items of look-ahead), side-band filter, 1:125"""
    import torch
    import qradiolink_amd as q
    n = 5 * 1024 + 700
    t = np.arange(n) / 8000.0
    audio = np.stack([0.9 * np.sin(2 * np.pi * 700 * t) + 0.5 * np.sin(2 * np.pi * 1500 * t),      # loud enough for the clipper to act
                      np.random.default_rng(4).uniform(-1.2, 1.2, n)]).astype(np.float32)
    mod = q.AMod(qrl_ctx, modem, batch=2, max_samples=min(chunk, n), bb_gain=0.8)
    parts = [mod.process(torch.from_numpy(np.ascontiguousarray(audio[:, s:s + chunk])).cuda()).cpu().numpy() for s in range(0, n, chunk)]
    mod.close()
    got = np.concatenate(parts, axis=1)
    assert got.shape == (2, 125 * 5 * 1024)
    for b in range(2):
        want = orc.mod_ssb(audio[b], sb=sb, bb_gain=0.8)
        g, w = got[b].view(np.float32) + np.float32(0), want.view(np.float32) + np.float32(0)
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), "stream %d differs" % b


def test_ssb_voice_loopback_on_gpu(qrl_ctx):
    import torch
    import qradiolink_amd as q
    n = 8192 + 2
    audio = (0.4 * np.sin(2 * np.pi * 700 * np.arange(n) / 8000.0)).astype(np.float32)
    mod = q.AMod(qrl_ctx, q.MODEM_LSB2500, batch=1, max_samples=n)
    iq = mod.process(torch.from_numpy(audio[None, :]).cuda()) * 0.5
    mod.close()
    assert iq.shape[1] == 125 * 8192
    rms = {}
    for modem in (q.MODEM_LSB2500, q.MODEM_USB2500):
        dem = q.Demod(qrl_ctx, modem, batch=1, max_chunk=iq.shape[1])
        out = q.collect(dem, iq.contiguous(), iq.shape[1])
        dem.close()
        a = out["audio"][0][2048:6144].astype(np.float64)
        rms[modem] = np.sqrt(np.mean(a ** 2)) if a.size else 0.0      # (the other side band may even fall under the -140 dB gate)
        if modem == q.MODEM_LSB2500:
            spec = np.abs(np.fft.rfft(a * np.hanning(a.size)))
            assert abs(np.argmax(spec) * 8000.0 / a.size - 700.0) < 3.0
    assert rms[q.MODEM_USB2500] < 0.02 * rms[q.MODEM_LSB2500]


# ---- M17 modulator (gr_mod_m17): raw dibits, 2500 samples per 3 bytes
@pytest.mark.parametrize("chunk", [96, 48, 3, 45])
def test_m17_modulator_bit_exact(qrl_ctx, chunk):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(6)
    data = rng.integers(0, 256, (2, 96), dtype=np.uint8)
    mod = q.Mod(qrl_ctx, q.MODEM_M17, batch=2, max_bytes=96, bb_gain=0.9)
    assert mod.spb == 0 and mod.spblock == 2500 and mod.bytes_per_block == 3
    parts = [mod.process(torch.from_numpy(np.ascontiguousarray(data[:, s:s + chunk])).cuda()).cpu().numpy() for s in range(0, 96, chunk)]
    got = np.concatenate(parts, axis=1)
    with pytest.raises(q.QrlError):
        mod.process(torch.from_numpy(data[:, :4].copy()).cuda())            # not a multiple of 3 bytes
    mod.close()
    assert got.shape == (2, 80000)
    for b in range(2):
        want = orc.mod_m17(data[b], bb_gain=0.9)
        g, w = got[b].view(np.float32) + np.float32(0), want.view(np.float32) + np.float32(0)
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), "stream %d differs" % b


def test_m17_tx_rx_frame_loopback_on_gpu(qrl_ctx):
    """bytes -> gr_mod_m17 -> gr_demod_m17 -> frame synchroniser: the bits that went in come out (device TX, device RX)"""
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(7)
    data = rng.integers(0, 256, 144, dtype=np.uint8)
    mod = q.Mod(qrl_ctx, q.MODEM_M17, batch=1, max_bytes=144)
    iq = (mod.process(torch.from_numpy(data[None, :]).cuda()) * 0.05).contiguous()
    mod.close()
    dem = q.Demod(qrl_ctx, q.MODEM_M17, batch=1, max_chunk=iq.shape[1])
    out = q.collect(dem, iq, iq.shape[1])
    dem.close()
    got = "".join(map(str, out["bits_a"][0]))
    want = "".join(map(str, np.unpackbits(data)[200:900]))
    assert want in got


# ---- DSSS "BPSK 8" modulator (gr_mod_dsss): 1 000 000 samples per byte
@pytest.mark.parametrize("cuts", [[4], [1, 3], [2, 1, 1]])
def test_dsss_modulator_bit_exact(qrl_ctx, cuts):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(8)
    data = rng.integers(0, 256, (2, 4), dtype=np.uint8)
    mod = q.Mod(qrl_ctx, q.MODEM_BPSK8, batch=2, max_bytes=4, bb_gain=0.9)
    assert mod.spb == 1000000
    parts, pos = [], 0
    for c in cuts:
        parts.append(mod.process(torch.from_numpy(np.ascontiguousarray(data[:, pos:pos + c])).cuda()).cpu().numpy())
        pos += c
    mod.close()
    got = np.concatenate(parts, axis=1)
    assert got.shape == (2, 4000000)
    for b in range(2):
        want = orc.mod_dsss(data[b], bb_gain=0.9)
        g, w = got[b].view(np.float32) + np.float32(0), want.view(np.float32) + np.float32(0)
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), "stream %d differs" % b


def test_dsss_tx_rx_loopback_on_gpu(qrl_ctx):
    """30 bytes at 8 bit/s: gr_mod_dsss -> gr_demod_dsss on the device; the information bits come back on one Viterbi branch"""
    import torch
    import qradiolink_amd as q
    data = np.random.default_rng(3).integers(0, 256, 30, dtype=np.uint8)
    mod = q.Mod(qrl_ctx, q.MODEM_BPSK8, batch=1, max_bytes=30)
    iq = (mod.process(torch.from_numpy(data[None, :]).cuda()) * 0.05).contiguous()
    mod.close()
    dem = q.Demod(qrl_ctx, q.MODEM_BPSK8, batch=1, max_chunk=1 << 22)
    out = q.collect(dem, iq, 1 << 22)
    dem.close()
    want = "".join(map(str, np.unpackbits(data)[20:100]))
    assert any(want in "".join(map(str, out[p][0])) for p in ("bits_a", "bits_b"))


# ---- DMR modulator (gr_mod_dmr): raw dibits, 2500 samples per 3 bytes, gr_zero_idle_bursts(62) in the chain
@pytest.mark.parametrize("chunk", [264, 33, 3])
def test_dmr_modulator_bit_exact(qrl_ctx, chunk):
    """QRL_MODEM_DMR TX against the oracle's gr_mod_dmr chain (src/gr/gr_mod_dmr.cpp:26-90), in one call and in ragged calls, without and with
    "zero_samples" tags -- runs that straddle call boundaries, overlap (the later tag ends the earlier run), lie in the first 62 items (no
    item to match) or reach past the end"""
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(16)
    nb = 264                                          # 8 DMR bursts of 33 bytes: 5280 items at 24 ksps
    data = rng.integers(0, 256, (2, nb), dtype=np.uint8)
    tags = {0: [(30, 100), (1500, 720), (2000, 50), (2635, 12), (5200, 400)], 1: [(1439 + 62, 1320)]}
    for with_tags in (False, True):
        mod = q.Mod(qrl_ctx, q.MODEM_DMR, batch=2, max_bytes=nb, bb_gain=0.9)
        assert mod.spb == 0 and mod.spblock == 2500 and mod.bytes_per_block == 3
        if with_tags:
            mod.add_zero_runs([(s, t, c) for s, l in tags.items() for t, c in l])
        parts = [mod.process(torch.from_numpy(np.ascontiguousarray(data[:, s:s + chunk])).cuda()).cpu().numpy() for s in range(0, nb, chunk)]
        got = np.concatenate(parts, axis=1)
        mod.close()
        assert got.shape == (2, nb // 3 * 2500)
        for b in range(2):
            want = orc.mod_dmr(data[b], bb_gain=0.9, zero_runs=tags[b] if with_tags else None)
            g, w = got[b].view(np.float32) + np.float32(0), want.view(np.float32) + np.float32(0)
            assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), "stream %d differs (tags %s)" % (b, with_tags)
    # the tagged stretch is silence at the output: the tag at 1500 zeroes items 1438 ... until the tag at 2000 reloads the counter with 50 (item
    # 1938): the signal is back from item 1988 on (x 125 / 3 at 1 Msps; 60 items of margin for the interpolator's span)
    w = orc.mod_dmr(data[0], bb_gain=0.9, zero_runs=tags[0])
    lo, hi = (1438 + 60) * 125 // 3, (1988 - 60) * 125 // 3
    assert np.abs(w[lo:hi]).max() < 1e-3 and np.abs(w[hi + 6000:hi + 12000]).mean() > 0.3


def test_dmr_tx_rx_dibit_loopback_on_gpu(qrl_ctx):
    """bytes -> gr_mod_dmr -> gr_demod_dmr: the dibits that went in come out of the receiver's symbol port (device TX, device RX)"""
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(17)
    data = rng.integers(0, 256, 330, dtype=np.uint8)
    mod = q.Mod(qrl_ctx, q.MODEM_DMR, batch=1, max_bytes=330)
    iq = (mod.process(torch.from_numpy(data[None, :]).cuda()) * 0.05).contiguous()
    mod.close()
    dem = q.Demod(qrl_ctx, q.MODEM_DMR, batch=1, max_chunk=iq.shape[1])
    out = q.collect(dem, iq, iq.shape[1])
    dem.close()
    got = "".join(map(str, out["bits_a"][0]))
    want = "".join(map(str, np.unpackbits(data)[1200:1900]))   # (the chain is ~ 320 symbols late: gr_zero_idle_bursts alone 1439 items = 288 symbols)
    assert want in got


# ---- TX-side CTCSS of gr_mod_nbfm (set_ctcss: tone source + add_ff, band-pass audio filter, x0.85)
@pytest.mark.parametrize("chunk", [8000, 1000, 324])
def test_nbfm_modulator_ctcss_bit_exact(qrl_ctx, chunk):
    """qrl_amod_set_ctcss against the oracle's gr_mod_nbfm with set_ctcss(tone) (src/gr/gr_mod_nbfm.cpp:101-135): 88.5 Hz and the off-table
    tone 81.5 Hz of the reference's tone list (src/ext/utils.h:17), in one call and in ragged calls; then set_ctcss(0) after it had been on:
    the low-pass again, but _audio_amplify 0.98"""
    import torch
    import qradiolink_amd as q
    n = 8000
    audio = np.stack([0.5 * np.sin(2 * np.pi * 700 * np.arange(n) / 8000.0), np.random.default_rng(31).uniform(-0.7, 0.7, n)]).astype(np.float32)
    for tone, oracle_arg in ((88.5, 88.5), (81.5, 81.5), (0.0, -1.0)):
        mod = q.AMod(qrl_ctx, q.MODEM_NBFM5000, batch=2, max_samples=chunk, bb_gain=0.75)
        mod.set_ctcss(88.5)
        mod.set_ctcss(tone)
        parts = [mod.process(torch.from_numpy(np.ascontiguousarray(audio[:, s:s + chunk])).cuda()).cpu().numpy() for s in range(0, n, chunk)]
        mod.close()
        got = np.concatenate(parts, axis=1)
        for b in range(2):
            want = orc.mod_nbfm(audio[b], filter_width=5000, bb_gain=0.75, ctcss=oracle_arg)
            g, w = got[b].view(np.float32) + np.float32(0), want.view(np.float32) + np.float32(0)
            assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), "tone %s stream %d differs" % (tone, b)
    with pytest.raises(q.QrlError):
        m = q.AMod(qrl_ctx, q.MODEM_USB2500, batch=1, max_samples=2048)
        try:
            m.set_ctcss(88.5)
        finally:
            m.close()


def test_nbfm_ctcss_tx_opens_the_ctcss_squelch_of_the_receiver_on_gpu(qrl_ctx):
    """gr_mod_nbfm with set_ctcss(88.5) -> gr_demod_nbfm with set_ctcss(88.5), both on the device: the receiver's tone squelch opens and the voice
    tone comes out; the same transmission WITHOUT the sub-tone (or with another one, 123.0 Hz) keeps the audio path shut"""
    import torch
    import qradiolink_amd as q
    n = 24000
    audio = (0.5 * np.sin(2 * np.pi * 700 * np.arange(n) / 8000.0)).astype(np.float32)
    rms = {}
    for tx_tone in (88.5, 0.0, 123.0):
        mod = q.AMod(qrl_ctx, q.MODEM_NBFM5000, batch=1, max_samples=n)
        if tx_tone:
            mod.set_ctcss(tx_tone)
        iq = (mod.process(torch.from_numpy(audio[None, :]).cuda()) * 0.05).contiguous()
        mod.close()
        dem = q.Demod(qrl_ctx, q.MODEM_NBFM5000, batch=1, max_chunk=iq.shape[1])
        dem.set_ctcss(88.5)
        out = q.collect(dem, iq, iq.shape[1])
        dem.close()
        a = out["audio"][0].astype(np.float64)
        rms[tx_tone] = float(np.sqrt(np.mean(a[-8000:] ** 2))) if a.size >= 8000 else 0.0
        if tx_tone == 88.5:
            seg = a[-8000:]
            spec = np.abs(np.fft.rfft(seg * np.hanning(seg.size)))
            assert abs(np.argmax(spec) * 8000.0 / seg.size - 700.0) < 3.0
    assert rms[88.5] > 0.2 and rms[0.0] < 1e-3 and rms[123.0] < 1e-3, rms


# ---- gr_mod_base::set_filter_width (src/gr/gr_mod_base.cpp:878-905): the analogue modulators' own set_filter_width designs
@pytest.mark.parametrize("modem,kind,fw,w", [(9, "nbfm", 5000, 4000), (8, "nbfm", 2500, 3000), (14, "am", 5000, 4000), (11, "usb", 2700, 2400), (12, "lsb", 2700, 2400)])
@pytest.mark.parametrize("chunk", [1 << 14, 1000])
def test_analog_modulator_set_filter_width_bit_exact(qrl_ctx, modem, kind, fw, w, chunk):
    """qrl_amod_set_filter_width against the oracle's chain with the reference setter's designs (pinned by tests/test_ref_chains.py::
    test_set_filter_width_of_the_analogue_blocks): called after some audio has gone through (the chain restarts), one call and ragged calls"""
    import torch
    import qradiolink_amd as q
    n = (5 * 1024 + 700) if kind in ("usb", "lsb") else 2000
    t = np.arange(n) / 8000.0
    audio = np.stack([0.6 * np.sin(2 * np.pi * 700 * t) + 0.3 * np.sin(2 * np.pi * 1500 * t), np.random.default_rng(41).uniform(-0.8, 0.8, n)]).astype(np.float32)
    c = min(chunk, n) // 4 * 4
    mod = q.AMod(qrl_ctx, modem, batch=2, max_samples=c, bb_gain=0.75)
    mod.process(torch.from_numpy(np.ascontiguousarray(audio[:, :c])).cuda())                 # something in flight: the setter restarts the chain
    mod.set_filter_width(w)
    parts = []
    for s in range(0, n, c):
        blk = audio[:, s:s + c]
        if kind == "nbfm" and blk.shape[1] % 4:
            blk = blk[:, :blk.shape[1] // 4 * 4]
        parts.append(mod.process(torch.from_numpy(np.ascontiguousarray(blk)).cuda()).cpu().numpy())
    got = np.concatenate(parts, axis=1)
    for b in range(2):
        a = audio[b] if kind != "nbfm" else audio[b][:n // 4 * 4]
        want = {"nbfm": lambda: orc.mod_nbfm(a, filter_width=fw, bb_gain=0.75, set_width=w),
                "am": lambda: orc.mod_am(a, filter_width=fw, bb_gain=0.75, set_width=w),
                "usb": lambda: orc.mod_ssb(a, sb=0, filter_width=fw, bb_gain=0.75, set_width=w),
                "lsb": lambda: orc.mod_ssb(a, sb=1, filter_width=fw, bb_gain=0.75, set_width=w)}[kind]()
        g = got[b][:want.size].view(np.float32) + np.float32(0)
        assert got.shape[1] >= want.size and want.size > 0
        assert np.array_equal(g.view(np.uint32), (want.view(np.float32) + np.float32(0)).view(np.uint32)), "stream %d differs" % b
        # and it is not the constructor's chain with that width (the setter's transition widths differ), except for AM where the setter repeats the constructor
        ctor = {"nbfm": lambda: orc.mod_nbfm(a, filter_width=w, bb_gain=0.75), "am": lambda: orc.mod_am(a, filter_width=w, bb_gain=0.75),
                "usb": lambda: orc.mod_ssb(a, sb=0, filter_width=w, bb_gain=0.75), "lsb": lambda: orc.mod_ssb(a, sb=1, filter_width=w, bb_gain=0.75)}[kind]()
        assert (kind == "am") == (ctor.size == want.size and np.array_equal(ctor, want))
    with pytest.raises(q.QrlError):
        mod.set_filter_width(100)                                                             # filters would not fit / below the SSB band edge
    mod.close()


# ---- the gr_mod_base back end behind the analogue modulators (qrl_amod_config.device_samp_rate / carrier_offset_hz)
@pytest.mark.parametrize("modem,kind,rate,offset", [(9, "nbfm", 4000000, 25000.0), (14, "am", 1000000, -12500.0), (11, "usb", 2000000, 0.0), (12, "lsb", 10000000, 50000.0)])
def test_analog_modulator_back_end_bit_exact(qrl_ctx, modem, kind, rate, offset):
    """rotator at 1 Msps + interpolation to the device rate behind NBFM / AM / SSB, in ragged calls, with a phase-continuous retune in the middle
    (gr_mod_base.cpp:38,215-258,799-805) -- the same oracle back end as behind the digital modulators"""
    import torch
    import qradiolink_amd as q
    n = (4 * 1024 + 700) if kind in ("usb", "lsb") else 1200
    t = np.arange(n) / 8000.0
    audio = np.stack([0.6 * np.sin(2 * np.pi * 700 * t) + 0.3 * np.sin(2 * np.pi * 1500 * t), np.random.default_rng(43).uniform(-0.8, 0.8, n)]).astype(np.float32)
    cuts = [n] if kind == "lsb" else [400, 4, 796] if kind != "usb" else [1500, 1024, n - 2524]
    mod = q.AMod(qrl_ctx, modem, batch=2, max_samples=max(cuts), bb_gain=0.75, device_samp_rate=rate, carrier_offset_hz=offset)
    assert mod.spa == 125 * (rate // 1000000)
    parts, pos, counts = [], 0, []
    for i, c in enumerate(cuts):
        if i == 2 and offset != 0.0:
            mod.set_carrier_offset(-2 * offset)
        parts.append(mod.process(torch.from_numpy(np.ascontiguousarray(audio[:, pos:pos + c])).cuda()).cpu().numpy())
        counts.append(parts[-1].shape[1] // (rate // 1000000))
        pos += c
    mod.close()
    got = np.concatenate(parts, axis=1)
    for b in range(2):
        x1 = {"nbfm": lambda: orc.mod_nbfm(audio[b], filter_width=5000, bb_gain=0.75), "am": lambda: orc.mod_am(audio[b], bb_gain=0.75),
              "usb": lambda: orc.mod_ssb(audio[b], sb=0, bb_gain=0.75), "lsb": lambda: orc.mod_ssb(audio[b], sb=1, bb_gain=0.75)}[kind]()
        inc0 = orc.phase_inc_to_turn(2 * np.pi * offset / 1e6)
        if len(cuts) == 3 and offset != 0.0:
            k = counts[0] + counts[1]                                    # 1 Msps samples produced before the retune
            inc1 = orc.phase_inc_to_turn(2 * np.pi * -2 * offset / 1e6)
            rot = np.concatenate([orc.rotator(x1[:k], inc0), orc.rotator(x1[k:], inc1, (k * inc0) & (2 ** 64 - 1))])
        else:
            rot = orc.rotator(x1, inc0)
        ref = orc.tx_interp(rot, rate) if rate != 1000000 else rot
        assert got[b].size == ref.size, (got[b].size, ref.size)
        assert np.array_equal((got[b].view(np.float32) + np.float32(0)).view(np.uint32), (ref.view(np.float32) + np.float32(0)).view(np.uint32)), "stream %d differs" % b
    m1 = q.AMod(qrl_ctx, q.MODEM_NBFM5000, batch=1, max_samples=64)
    with pytest.raises(q.QrlError):
        m1.set_carrier_offset(1000.0)                                    # created without the back end
    m1.close()


# ---- CW600USB: the SSB chain fed by the key's tone source (gr_mod_base.cpp:144,180,679-683,948-956)
def _cw_reference(segments, set_width=0):
    """segments: [(n samples, key down)] -> the oracle's gr_mod_ssb(125, 1e6, ., 1000, 0) over sig_source_f(8000, GR_SIN_WAVE, 600, 0.001 | 0.98, 1)"""
    tone, k0 = [], 0
    for n, down in segments:
        tone.append(orc.sig_source_sin(8000, 600, 0.98 if down else 0.001, n, k0=k0, offset=1.0))
        k0 += n
    return orc.mod_ssb(np.concatenate(tone), sb=0, filter_width=1000, set_width=set_width)


def test_cw_modulator_bit_exact(qrl_ctx):
    """qrl_amod QRL_MODEM_CW600USB + qrl_amod_set_cw_k against the oracle: keyed in the middle of the stream (the amplitude changes with the next call, the
    tone's phase runs on), ragged calls; then the same through qrl_amod_set_filter_width(800)"""
    import qradiolink_amd as q
    segs = [(1500, False), (2500, True), (1024, True), (3000, False), (168, True)]
    for width in (0, 800):
        mod = q.AMod(qrl_ctx, q.MODEM_CW600USB, batch=2, max_samples=3000)
        if width:
            mod.set_filter_width(width)
        parts = []
        for n, down in segs:
            mod.set_cw_k(down)
            parts.append(mod.process_cw(n).cpu().numpy())
        got = np.concatenate(parts, axis=1)
        want = _cw_reference(segs, set_width=width)
        assert got.shape[1] == want.size == 125 * 1024 * ((sum(n for n, _ in segs) - 2) // 1024)
        for b in range(2):
            assert np.array_equal((got[b].view(np.float32) + np.float32(0)).view(np.uint32), (want.view(np.float32) + np.float32(0)).view(np.uint32)), (width, b)
        mod.close()
    # qrl_amod_set_filter_width in the MIDDLE of a transmission: the chain restarts, the key's sig_source_f keeps running -- what follows
    # equals the oracle's chain (setter designs) over the tone from sample 1500 on
    mod = q.AMod(qrl_ctx, q.MODEM_CW600USB, batch=2, max_samples=3000)
    mod.set_cw_k(True)
    mod.process_cw(1500)
    mod.set_filter_width(800)
    got = np.concatenate([mod.process_cw(n).cpu().numpy() for n in (2500, 1024, 168)], axis=1)
    want = orc.mod_ssb(orc.sig_source_sin(8000, 600, 0.98, 2500 + 1024 + 168, k0=1500, offset=1.0), sb=0, filter_width=1000, set_width=800)
    assert got.shape[1] == want.size > 0
    for b in range(2):
        assert np.array_equal((got[b].view(np.float32) + np.float32(0)).view(np.uint32), (want.view(np.float32) + np.float32(0)).view(np.uint32)), b
    mod.close()
    # key down: a carrier 600 Hz above the suppressed carrier (the cessb stretcher holds a full-scale tone at ~ 0.09); key up: the 0.001 tone at the chain's
    # small-signal gain 0.42, > 40 dB below
    x = _cw_reference([(8192, True)])
    y = _cw_reference([(8192, False)])
    assert 0.05 < np.abs(x[400000:800000]).mean() < 0.2 and np.abs(y[400000:800000]).mean() < 1e-3
    spec = np.abs(np.fft.fft(x[400000:800000] * np.hanning(400000)))
    assert abs(np.fft.fftfreq(400000, 1e-6)[np.argmax(spec)] - 600.0) < 5.0
    m = q.AMod(qrl_ctx, q.MODEM_USB2500, batch=1, max_samples=2048)
    with pytest.raises(q.QrlError):
        m.set_cw_k(True)
    m.close()
