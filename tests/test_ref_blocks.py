"""PINS parts of the oracle against the REAL reference: oracle/_ref/libqrl_ref.so holds the reference's own GNU Radio custom blocks
(src/gr/gr_dmr_dmo_sink.cpp -- SURVEY a37b --, gr_deframer_bb.cpp -- 8(f) rank 1 --, gr_4fsk_discriminator.cpp, rssi_tag_block.cpp),
compiled unmodified from where they lie against oracle/gr_stub (a stand-in for the GNU Radio base classes; make -C oracle ref).  Their
work() functions are driven with scheduler-like ragged calls and compared with the oracle's restatement of the same block.  Skipped
where the library has not been built (no /root/reference and no prebuilt copy)."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
import sig
from test_dmo_sink import _burst

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libqrl_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libqrl_ref.so not built (make -C oracle ref needs /root/reference)")
P = lambda a: a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def ref():
    lib = C.CDLL(REF)
    lib.ref_dmo_sink.restype = C.c_size_t
    lib.ref_deframer.restype = C.c_size_t
    lib.ref_rssi_tag.restype = C.c_size_t
    return lib


def _ref_dmo(ref, x, chunk):
    x = np.ascontiguousarray(x, np.float32)
    rec = np.zeros(40 * 256, np.uint8)
    n = ref.ref_dmo_sink(P(x), C.c_size_t(x.size), C.c_size_t(chunk), P(rec), C.c_size_t(256))
    return [(int(rec[40 * i]), int(rec[40 * i + 1]), int(rec[40 * i + 2]), rec[40 * i + 4:40 * i + 37].tobytes()) for i in range(n)]


@pytest.mark.parametrize("seed,chunk", [(1, 1 << 20), (2, 997), (3, 4096), (4, 333)])
def test_dmo_slicer_oracle_equals_the_reference_block(ref, seed, chunk):
    """voice calls, data bursts, CSBKs, noise, offset and a different deviation; gaps of noise between the calls.
    The stream starts with 2000 samples of noise: the reference block never initialises its 1440-sample ring, its sign shift
    registers, m_control or the centre / threshold averages (gr_dmr_dmo_sink.cpp:29-46), and the 132nd symbol of a slot is read one
    sample AHEAD of the write pointer (:118-122), i.e. from the previous lap of the ring -- for a burst inside the first 1440
    samples of a fresh block that is uninitialised heap memory.  The oracle (and the kernel) define that state as zero; behind one
    lap of the ring the block's output no longer depends on it."""
    rng = np.random.default_rng(seed)
    parts = [np.zeros(2000, np.float32)]
    for k in range(4):
        kinds = [["lc", "voice_sync", "v", "v", "v", "v", "v", "voice_sync", "v", "v", "term"], ["csbk"], ["lc", "voice_sync", "v", "csbk"],
                 ["voice_sync", "v", "v", "v", "v", "v", "voice_sync"]][(k + seed) % 4]
        frames = _burst(rng, kinds, cc=int(rng.integers(0, 16)))
        x = sig.dmr_samples(frames, scale=float(rng.uniform(0.2, 0.6))) * float(rng.uniform(0.6, 1.2)) + float(rng.uniform(-0.1, 0.1))
        parts += [x, np.zeros(int(rng.integers(500, 4000)), np.float32)]
    x = np.concatenate(parts).astype(np.float32)
    x = x + 0.01 * rng.standard_normal(x.size).astype(np.float32)
    want = _ref_dmo(ref, x, chunk)
    got = orc.DmoSink().process(x, cap=256)
    assert len(want) >= 12
    assert got == want


def test_dmo_slicer_on_pure_noise_and_random_levels(ref):
    rng = np.random.default_rng(9)
    x = rng.choice(np.array([-0.3, -0.1, 0.1, 0.3], np.float32), 120000) + 0.02 * rng.standard_normal(120000).astype(np.float32)
    x = np.repeat(x[:24000], 5)[:120000].astype(np.float32)        # symbol-shaped random levels: false syncs are possible
    x[:2000] = 0.02 * rng.standard_normal(2000).astype(np.float32)  # one lap of the ring before anything can be sliced (see above)
    assert orc.DmoSink().process(x, cap=256) == _ref_dmo(ref, x, 1111)


@pytest.mark.parametrize("type_", [1, 2, 3])
def test_deframer_oracle_equals_the_reference_block(ref, type_):
    rng = np.random.default_rng(type_)
    sync = {1: [0xED, 0x89], 2: [0xB5], 3: [0xED, 0x89]}[type_]
    bits = []
    for k in range(40):
        bits += list(rng.integers(0, 2, int(rng.integers(0, 60))))
        word = np.unpackbits(np.array(sync, np.uint8))
        if k % 5 == 4:
            word = word.copy(); word[int(rng.integers(0, word.size))] ^= 1      # a damaged sync word
        bits += list(word) + list(rng.integers(0, 2, {1: 64, 2: 32, 3: 384}[type_]))
    bits = np.array(bits, np.uint8)
    out = np.zeros(2 * bits.size + 64, np.uint8)
    for chunk in (1 << 20, 257, 31):
        n = ref.ref_deframer(type_, P(bits), C.c_size_t(bits.size), C.c_size_t(chunk), P(out), C.c_size_t(out.size))
        st = np.zeros(3, np.uint32)
        got = np.concatenate([orc.deframer(type_, bits[s:s + chunk], st) for s in range(0, bits.size, chunk)])
        assert n > 100 and np.array_equal(got, out[:n]), (type_, chunk)


def test_4fsk_discriminator_and_rssi_tag_equal_the_reference_blocks(ref):
    rng = np.random.default_rng(4)
    m = rng.standard_normal((4, 5000)).astype(np.float32)
    m[:, ::50] = m[0, ::50]                                          # ties -> 0 + 0j
    out = np.zeros(5000, np.complex64)
    ref.ref_4fsk_discriminator(P(m[0]), P(m[1]), P(m[2]), P(m[3]), C.c_size_t(5000), P(out))
    k = np.argmax(m, axis=0)
    strict = np.array([np.sum(m[:, i] == m[k[i], i]) == 1 for i in range(5000)])
    pts = np.array([-0.707107 - 0.707107j, -0.707107 + 0.707107j, 0.707107 + 0.707107j, 0.707107 - 0.707107j], np.complex64)
    assert np.array_equal(out, np.where(strict, pts[k], 0).astype(np.complex64))
    x = ((rng.standard_normal(10000) + 1j * rng.standard_normal(10000)) * 0.05).astype(np.complex64)
    y = np.zeros_like(x)
    off, db = np.zeros(64, np.uint64), np.zeros(64, np.float32)
    for chunk in (1 << 20, 777):
        n = ref.ref_rssi_tag(P(x), C.c_size_t(x.size), C.c_size_t(chunk), P(y), P(off), P(db), C.c_size_t(64))
        want = orc.rssi_tag(x, 0.0)
        assert n == want.size == 33 and np.array_equal(y, x)
        assert np.allclose(db[:n], want, rtol=0, atol=2e-5)           # (the block uses libm log10f; the oracle's log is the deterministic one)
        assert np.array_equal(off[:n], 299 + 300 * np.arange(n))


def test_deemphasis_taps_equal_the_reference_function(ref):
    for fs in (8000, 20000, 200000):
        a, b = (C.c_double * 2)(), (C.c_double * 2)()
        ref.ref_deemph_taps(fs, C.c_double(50e-6), a, b)
        oa, ob = orc.deemph_taps(fs, 50e-6)
        assert list(a) == oa and list(b) == ob          # bit-identical doubles
    a, b = (C.c_double * 2)(), (C.c_double * 2)()
    ref.ref_preemph_taps(8000, C.c_double(50e-6), a, b)      # gr_mod_nbfm's pre-emphasis (default upper corner)
    oa, ob = orc.preemph_taps(8000, 50e-6)
    assert list(a) == oa and list(b) == ob


def test_zero_idle_bursts_oracle_equals_the_reference_block(ref):
    """one down-counter per stream, (re)loaded by the tag at an item's offset: a run that starts inside another ends it there"""
    rng = np.random.default_rng(6)
    n = 20000
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    offs = np.array([100, 900, 1000, 5000, 5010, 7200, 7920, 15000, 19990], np.uint64)
    cnts = np.array([720, 720, 50, 720, 0, 720, 720, 1, 100], np.uint64)         # overlapping, zero-length and clipped runs
    runs = np.stack([np.zeros_like(offs), offs, cnts], axis=1).astype(np.uint64)
    want = np.zeros_like(x)
    orc.lib.orc_zero_idle_bursts(P(x), C.c_size_t(n), P(np.ascontiguousarray(runs)), C.c_size_t(offs.size), P(want))
    for chunk in (1 << 20, 720, 333):
        got = np.zeros_like(x)
        ref.ref_zero_idle_bursts(P(x), C.c_size_t(n), C.c_size_t(chunk), P(offs), P(cnts), C.c_size_t(offs.size), P(got))
        assert np.array_equal(got, want)
    assert np.count_nonzero(want == 0) == 720 + (100 + 50) + 10 + (720 + 720) + 1 + 10   # [100,820) [900,1000)->[1000,1050) [5000,5010) ... [19990,20000)


def test_zero_idle_bursts_with_delay_oracle_equals_the_reference_block(ref):
    """gr_zero_idle_bursts(delay = 62), the form gr_mod_dmr builds (src/gr/gr_mod_dmr.cpp:57,63): the history of 2 x 720 items delays the stream
    by 1439 items, a tag at offset T zeroes the OUTPUT items T - 62 ...; a tag in the first 62 items has no item to match.  One work() call over
    the whole stream (across calls the reference misses tags within the first `delay` items of a call's window: not restated)."""
    rng = np.random.default_rng(7)
    n = 12000
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    offs = np.array([30, 62, 500, 1200, 1230, 4000, 11990], np.uint64)
    cnts = np.array([10, 5, 720, 100, 3, 1320, 500], np.uint64)
    runs = np.stack([np.zeros_like(offs), offs, cnts], axis=1).astype(np.uint64)
    for delay in (62, 0):
        want, got = np.zeros_like(x), np.zeros_like(x)
        orc.lib.orc_zero_idle_bursts_delay(P(x), C.c_size_t(n), C.c_uint(delay), P(np.ascontiguousarray(runs)), C.c_size_t(offs.size), P(want))
        ref.ref_zero_idle_bursts_delay(P(x), C.c_size_t(n), C.c_uint(delay), P(offs), P(cnts), C.c_size_t(offs.size), P(got))
        assert np.array_equal(got, want), delay
    # delay 62: the stream itself is 1439 items late; the tag at 62 zeroes items 0..4 (already zero), the one at 4000 items 3938..5257
    orc.lib.orc_zero_idle_bursts_delay(P(x), C.c_size_t(n), C.c_uint(62), P(np.ascontiguousarray(runs)), C.c_size_t(offs.size), P(want))
    assert np.array_equal(want[5258:5300], x[5258 - 1439:5300 - 1439]) and not want[3938:5258].any() and want[3937] == x[3937 - 1439]


def test_dsss_decoder_oracle_equals_the_reference_block(ref):
    """gr::dsss::dsss_decoder_cc_impl.cc itself (FIR kernel and RRC design supplied by gr_stub with the oracle's summation order):
    matched-filter taps, window positions (set_history(325): the windows of output I start at 325 (I - 2) + 1 + j), first-maximum
    search and scaling.  The block takes |v| with std::abs (hypotf), the oracle sqrtf(re^2 + im^2): a different pick between two
    candidates within an ulp of each other would be legitimate; none occurs on these inputs."""
    ref.ref_dsss_decoder.restype = C.c_size_t
    rng = np.random.default_rng(8)
    bits = rng.integers(0, 2, 60)
    code = np.array([1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1, 0, 1])
    chips = np.concatenate([code if b == 0 else 1 - code for b in bits]) * 2.0 - 1
    up = np.zeros(chips.size * 25)
    up[::25] = chips
    x = np.convolve(up, orc.root_raised_cosine(25, 25, 1, 0.35, 275).astype(np.float64))[:up.size]
    x = (x * np.exp(0.7j) + 0.05 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
    for n in (x.size, 5000, 600, 599, 274):
        xs = np.ascontiguousarray(x[:n])
        want_taps = np.zeros(600, np.float32)
        out = np.zeros(xs.size // 325 + 8, np.complex64)
        for per_call in (1 << 20, 1, 7):
            m = ref.ref_dsss_decoder(P(xs), C.c_size_t(n), 25, C.c_size_t(per_call), P(out), P(want_taps))
            got = orc.dsss_decoder(xs, 25)
            assert m == got.size, (n, m, got.size)
            assert np.array_equal(got.view(np.uint32), out[:m].view(np.uint32)), (n, per_call)
        assert np.array_equal(orc.dsss_taps(25).view(np.uint32), want_taps.view(np.uint32))
    assert orc.dsss_decoder(x, 25).size == (x.size - 275) // 325 + 1


def test_cessb_clipper_and_stretcher_oracle_equal_the_reference_blocks(ref):
    """src/gr/cessb/clipper_cc_impl.cc and stretcher_cc_impl.cc themselves (VOLK kernels in generic form from gr_stub, cos / sin =
    the shared deterministic polynomial): operation order of the polar clipper; the stretcher's five-point envelope window, its
    carry of two envelope values across chunks, its two items of look-ahead and its whole-chunk output count"""
    ref.ref_cessb_stretcher.restype = C.c_size_t
    orc.lib.orc_cessb_stretcher.restype = C.c_size_t
    rng = np.random.default_rng(12)
    x = ((rng.standard_normal(5 * 1024) + 1j * rng.standard_normal(5 * 1024)) * 0.5).astype(np.complex64)
    x[100:140] = 0                                                        # zero magnitude: atan2(0, 0), 0 / 1
    a, b = np.zeros_like(x), np.zeros_like(x)
    ref.ref_cessb_clipper(P(x), C.c_size_t(x.size), C.c_float(0.95), P(a))
    orc.lib.orc_cessb_clipper(P(x), C.c_size_t(x.size), C.c_float(0.95), P(b))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for n in (5 * 1024, 4 * 1024 + 2, 4 * 1024 + 1, 1025, 1026):
        xs = np.ascontiguousarray(a[:n])
        want_n = orc.lib.orc_cessb_stretcher(P(xs), C.c_size_t(n), P(b))
        for chunks in (8, 1, 2):
            got = np.zeros(n + 8, np.complex64)
            m = ref.ref_cessb_stretcher(P(xs), C.c_size_t(n), C.c_size_t(chunks), P(got))
            assert m == want_n == 1024 * ((n - 2) // 1024)
            assert np.array_equal(got[:m].view(np.uint32), b[:m].view(np.uint32)), (n, chunks)


def test_dsss_encoder_chips_equal_the_reference_block(ref):
    """gr::dsss::dsss_encoder_bb: bit 0 -> the Barker-13 code, bit 1 -> its complement, MSB first; what k_tx_spread / orc_mod_dsss do
    with the coded bits (checked through the chips' signs in the oracle's modulator input)"""
    ref.ref_dsss_encoder.restype = C.c_size_t
    data = np.random.default_rng(13).integers(0, 256, 40, dtype=np.uint8)
    out = np.zeros(40 * 104, np.uint8)
    assert ref.ref_dsss_encoder(P(data), C.c_size_t(40), P(out)) == 40 * 104
    code = np.array([1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1, 0, 1], np.uint8)
    want = np.concatenate([code if b == 0 else 1 - code for b in np.unpackbits(data)])
    assert np.array_equal(out, want)
