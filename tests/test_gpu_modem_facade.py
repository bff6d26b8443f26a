"""The gr_modem / gr_demod_base / gr_mod_base-shaped C++ facade (qradiolink_amd/host/gr_modem_hip.*) in a full TX -> RX loopback on
the GPU, driven by tests/host/test_modem.cpp the way radiocontroller.cpp drives gr_modem: per stream startTransmission, voice
frames, a text message, endTransmission; ragged asynchronous work() calls; demodulate() polled per stream.  Checked: callsign,
every voice payload in order, the text, the end-of-stream events, the two Viterbi branches (GMSK) and the modem_sync >= 16 voice
gate of the 1k modes (reference src/gr_modem.cpp:1067-1090, 1338)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host", "test_modem")

pytestmark = pytest.mark.gpu


def _events(tmp_path, mode, streams, frames, host_loop=False):
    """host_loop = False: the default boundary (frame synchroniser on the device, demodulate() walks framed records);
    True: the reference's per-bit loop on the host (QRL_TEST_HOSTLOOP, the checker)"""
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])
    out = tmp_path / ("events_host.txt" if host_loop else "events.txt")
    env = dict(os.environ)
    env.pop("QRL_TEST_HOSTLOOP", None)
    if host_loop:
        env["QRL_TEST_HOSTLOOP"] = "1"
    r = subprocess.run([EXE, "loopback", str(mode), str(streams), str(frames), str(out)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    ev = {s: [] for s in range(streams)}
    for line in out.read_text().splitlines():
        s, kind, *rest = line.split(" ", 2)
        ev[int(s)].append((kind, rest[0] if rest else ""))
    assert dict(ev[0])["framing"] == ("host" if host_loop else "device")
    return ev


@pytest.mark.parametrize("mode,streams,frames", [(22, 3, 12), (26, 2, 3), (18, 2, 40), (7, 2, 10), (27, 2, 3)])
def test_device_framing_equals_the_host_loop(tmp_path, mode, streams, frames):
    """The boundary a maintainer binds: gr_modem_hip::demodulate with the L1 frame synchroniser on the device (qrl_framesync_* behind
    the demodulator, records { type, nbytes | _modem_sync << 16, payload } over PCIe, no per-bit host loop) delivers exactly the
    events of the same run with the reference's loop on the host (src/gr_modem.cpp:1119-1282 restated in gr_modem_hip::synchronize):
    same kinds, same payloads, same order per stream -- including the _modem_sync >= 16 voice gate of the 1k modes, which the host
    evaluates from the record."""
    dev = _events(tmp_path, mode, streams, frames)
    host = _events(tmp_path, mode, streams, frames, host_loop=True)
    skip = {"framing", "modem_sync", "rssi", "rssi_max", "spectra", "peak_bin", "peak_db", "scope_items", "scope_reads", "scope_max", "scope_power", "samples_in"}
    for s in range(streams):
        a = [e for e in dev[s] if e[0] not in skip]
        b = [e for e in host[s] if e[0] not in skip]
        assert a == b, s
    assert sum(len([e for e in dev[s] if e[0] not in skip]) for s in range(streams)) >= streams


def test_device_framing_host_time(tmp_path):
    """host CPU time of the RX boundary per work() call at 4096 streams of QPSK-250k (16384 samples = 4096 decoded bits per stream and
    call): the per-bit loop on the host against the frame synchroniser on the device; the factor is printed (pytest -s) and must be
    well above 1 for the polls"""
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])
    r = subprocess.run([EXE, "hosttime", "26", "4096", "12"], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    f = float(r.stdout.split("poll factor ")[1].split(",")[0])
    assert f > 2.0, r.stdout


def _payload(s, f, L):
    return bytes((17 * s + 31 * f + 7 * i + 1) & 0xFF for i in range(L)).hex()


@pytest.mark.parametrize("mode,L,streams,frames", [(22, 47, 3, 12),     # GMSK 10k: two Viterbi branches, FrameTypeVoice2 + reserved byte
                                                   (26, 1516, 2, 3),    # QPSK 250k: single branch, 24-bit sync words only
                                                   (18, 4, 2, 40)])     # 2FSK 1k: FrameTypeVoice1, the modem_sync >= 16 voice gate
def test_facade_loopback(tmp_path, mode, L, streams, frames):
    ev = _events(tmp_path, mode, streams, frames)
    for s in range(streams):
        kinds = [k for k, _ in ev[s]]
        if mode == 22:   # (the fast modes only know the IP / video / end sync words, the 1k modes only FrameTypeVoice1: gr_modem.cpp:1183-1282)
            assert ("callsign", "CALL%d" % s) in ev[s]
        audio = [v for k, v in ev[s] if k == "audio"]
        want = [_payload(s, f, L) for f in range(frames)]
        if mode == 26:
            assert audio == [] and "receiveend" in kinds      # voice frames are not a QPSK-250k frame type; the end frame is
            continue
        if mode == 18:
            # 1k modes: voice is only handed on once two syncs have been seen in close succession (modem_sync >= 16): the first
            # frame(s) after the callsign frame may be withheld, everything after them arrives in order
            # (+8 per sync word, -1 per searched bit: the gate opens at the 9th frame), everything after them arrives in order
            k = want.index(audio[0])
            assert k <= 10 and audio[:frames - k] == want[k:]   # (an 8-bit sync word also turns up inside the text / end frames later on)
            assert "callsign" not in kinds and "receiveend" not in kinds
            continue
        else:
            # every frame, in order; the misaligned Viterbi branch and the silence after the end frame decode to arbitrary bits, in
            # which a false 16-bit sync word is legitimate (about 2^-16 per bit): a few extra frames are tolerated
            it = iter(audio)
            assert all(w in it for w in want) and len(audio) <= frames + 3
            text = bytes.fromhex("".join(v for k, v in ev[s] if k == "text")).decode()
            assert text.startswith("hello from stream %d" % s)
        assert kinds.count("receiveend") >= 1 and kinds.count("endaudio") >= 1
        assert int(dict(ev[s])["modem_sync"]) >= 0


def test_facade_side_outputs(tmp_path):
    """enable_rssi / calibrate_rssi / get_rssi and enable_gui_fft / set_fft_size / get_FFT_data of the facade (gr_demod_base.cpp:978-986,
    1105-1113, 1227-1237, 1413-1418): the loopback signal (amplitude ~0.03 at 1 Msps) gives a finite RSSI per stream and spectra
    whose peak sits in the signal's band around DC (bin N/2 after the half swap)."""
    ev = _events(tmp_path, 22, 2, 3)
    d0 = dict(ev[0])
    for s in range(2):
        # while the burst is on the air: 2000 x |0.05 x 0.6..1|^2 ~ 1..5 -> 0..7 dB, -30 dB calibration; after it the level falls
        assert -45.0 < float(dict(ev[s])["rssi_max"]) < -15.0 and float(dict(ev[s])["rssi"]) < float(dict(ev[s])["rssi_max"]) - 20.0
    assert int(d0["spectra"]) >= 2
    assert abs(int(d0["peak_bin"]) - 2048) < 80 and float(d0["peak_db"]) > -70.0
    # the time-domain scope tap (enable_time_domain / set_sample_window / get_sample_data, gr_demod_base.cpp:988-1018, 1115-1147): one
    # 100 ksps item per ten input samples (the last call's items were still waiting when the loop ended), windows of at most 4002
    # items (4001 made even), the burst's power in them (|0.05 x 0.6..1|^2 while it is on the air, ~0 in the silence)
    n_in, items = int(d0["samples_in"]), int(d0["scope_items"])
    assert n_in // 10 - 14000 <= items <= n_in // 10 + 1 and int(d0["scope_max"]) <= 2 * 4002 and int(d0["scope_max"]) % 4 == 0
    assert 1e-5 < float(d0["scope_power"]) < 3e-3


@pytest.mark.parametrize("mode,kind,fw", [(9, "nbfm", 5000), (14, "am", 5000), (10, "wbfm", 75000)])
def test_facade_analog_audio(tmp_path, mode, kind, fw):
    """toggleRxMode(NBFM / AM / WBFM) -> work() -> demodulateAnalog() -> pcmAudio, two radios on one handle; the audio every
    stream's slot received equals the oracle chain bit for bit (gr_modem.cpp:996-1017, gr_demod_base.cpp:968-976)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc
    import sig
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])
    n = 300000
    xs = [sig.make_analog(kind, n=n, seed=7)[0], sig.make_analog(kind, n=n, seed=8, gap=(40000, 240000))[0]]
    (tmp_path / "iq.bin").write_bytes(np.stack(xs).tobytes())
    r = subprocess.run([EXE, "analog", str(mode), "2", str(tmp_path / "iq.bin"), str(tmp_path / "a")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for s in range(2):
        got = np.fromfile(tmp_path / ("a%d.bin" % s), np.float32) + np.float32(0)
        want = orc.demod_analog(xs[s], kind, filter_width=fw)["audio"] + np.float32(0)
        # the audio sink hands out packets of 640 samples; what is left below one packet at the end stays in the mailbox
        assert got.size == want.size // 640 * 640 and got.size >= 640
        assert np.array_equal(got.view(np.uint32), want[:got.size].view(np.uint32))


REF = os.path.join(ROOT, "oracle", "_ref", "libqrl_ref.so")


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libqrl_ref.so not built")
@pytest.mark.parametrize("mode", [22, 26, 18, 7, 0, 27, 19])
def test_facade_tx_framing_equals_the_reference_gr_modem(tmp_path, mode):
    """startTransmission / transmitDigitalAudio / TextData / BinData / VideoData / NetData / sendCallsign / endTransmission: the bytes
    gr_modem_hip hands to the byte source equal what the REFERENCE's gr_modem (src/gr_modem.cpp compiled unmodified into oracle/_ref,
    Qt stubbed) hands to gr_mod_base::set_data for the same calls -- frame(), transmit(), the preamble / callsign / end frames and
    the toggleTxMode frame lengths (src/gr_modem.cpp:105-199, 628-744, 804-978)"""
    import ctypes as C
    import numpy as np
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])
    r = subprocess.run([EXE, "txpin", str(mode), str(tmp_path / "tx.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "tx.bin", np.uint8)
    L = C.CDLL(REF)
    vp = C.c_void_p
    L.ref_modem_new.restype = vp
    for name, args in dict(ref_modem_free=[vp], ref_modem_init_tx=[vp, C.c_int], ref_modem_tx_frame_length=[vp], ref_modem_start_tx=[vp, C.c_char_p],
                           ref_modem_end_tx=[vp, C.c_char_p], ref_modem_send_callsign=[vp, C.c_char_p], ref_modem_tx_audio=[vp, vp, C.c_int],
                           ref_modem_tx_video=[vp, vp, C.c_int], ref_modem_tx_net=[vp, vp, C.c_int], ref_modem_tx_text=[vp, vp, C.c_int, C.c_int],
                           ref_modem_tx_bin=[vp, vp, C.c_int, C.c_int], ref_modem_tx_take=[vp, vp, C.c_size_t]).items():
        getattr(L, name).argtypes = args
    L.ref_modem_tx_take.restype = C.c_size_t
    m = L.ref_modem_new()
    L.ref_modem_init_tx(m, mode)
    n = L.ref_modem_tx_frame_length(m)
    arr = lambda f: np.array([f(i) & 0xFF for i in range(n)], np.uint8)
    L.ref_modem_start_tx(m, b"N0CALL")
    for f in range(3):
        a = arr(lambda i, f=f: 31 * f + 7 * i + 1)
        L.ref_modem_tx_audio(m, a.ctypes.data, n)
    text = b"the quick brown fox jumps over the lazy dog 0123456789"
    L.ref_modem_tx_text(m, text, len(text), 0x89EDAA)
    b = np.array([(200 - i) & 0xFF for i in range(2 * n + 3)], np.uint8)
    L.ref_modem_tx_bin(m, b.ctypes.data, b.size, 0xED77AA)
    a = arr(lambda i: i ^ 0x5A); L.ref_modem_tx_video(m, a.ctypes.data, n)
    a = arr(lambda i: i * 3); L.ref_modem_tx_net(m, a.ctypes.data, n)
    L.ref_modem_send_callsign(m, b"AB1CD")
    L.ref_modem_end_tx(m, b"N0CALL")
    buf = np.zeros(1 << 20, np.uint8)
    k = L.ref_modem_tx_take(m, buf.ctypes.data, buf.size)
    L.ref_modem_free(m)
    want = buf[:k]
    assert k > 10 * n and got.size == want.size, (got.size, k)
    assert np.array_equal(got, want)


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libqrl_ref.so not built")
@pytest.mark.parametrize("host_loop", [False, True])
@pytest.mark.parametrize("mode,streams,frames", [(22, 2, 10), (26, 2, 3), (18, 2, 30), (7, 2, 10)])
def test_facade_rx_events_equal_the_reference_gr_modem(tmp_path, mode, streams, frames, host_loop):
    """host_loop = False: the DEVICE frame synchroniser behind the facade (the default boundary); True: the per-bit loop on the host.
    The TX -> RX loopback again, with every bit vector of the demodulator's bit ports tapped: the same vectors go into the
    REFERENCE's gr_modem (src/gr_modem.cpp itself, oracle/_ref) call by call; its signals must be the facade's events, in order
    (synchronize / findSync / packBytes / processReceivedData).  Two-branch modes: the facade frames both Viterbi alignments with
    their own state, A then B; the reference side is therefore one gr_modem per branch, each fed its branch as
    the longer vector of the `>=` rule (src/gr_modem.cpp:1080-1090)."""
    import ctypes as C
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])
    env = dict(os.environ, QRL_TEST_TAP=str(tmp_path / "tap.txt"))
    env.pop("QRL_TEST_HOSTLOOP", None)
    if host_loop:
        env["QRL_TEST_HOSTLOOP"] = "1"
    r = subprocess.run([EXE, "loopback", str(mode), str(streams), str(frames), str(tmp_path / "ev.txt")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert ("0 framing " + ("host" if host_loop else "device")) in (tmp_path / "ev.txt").read_text()
    L = C.CDLL(REF)
    vp = C.c_void_p
    L.ref_modem_new.restype = vp
    for name, args in dict(ref_modem_free=[vp], ref_modem_init_rx=[vp, C.c_int], ref_modem_push=[vp, C.c_int, C.c_char_p, C.c_size_t],
                           ref_modem_demodulate=[vp], ref_modem_events=[vp, vp, C.c_size_t]).items():
        getattr(L, name).argtypes = args
    L.ref_modem_events.restype = C.c_size_t
    two = mode in (0, 15, 16, 17, 18, 19, 20, 21, 22, 24, 25)
    ms = [[L.ref_modem_new() for _ in range(2 if two else 1)] for _ in range(streams)]
    for per in ms:
        for m in per:
            L.ref_modem_init_rx(m, mode)
    facade = [[] for _ in range(streams)]
    reference = [[] for _ in range(streams)]
    buf = C.create_string_buffer(1 << 22)
    state = {"pending": None}        # (stream, {nr: bits}) of the demodulate() call being replayed

    returns = {"facade": [], "reference": []}

    def run(m, s):
        r = L.ref_modem_demodulate(m)
        n = L.ref_modem_events(m, buf, len(buf))
        reference[s] += buf.raw[:n].decode("latin-1").splitlines()
        state["ret"] = state.get("ret", 0) | (1 if r else 0)

    def flush():
        if state["pending"] is None:
            return
        s, vec = state["pending"]
        state["pending"] = None
        if two and len(vec) == 2:
            for k, nr in enumerate((1, 2)):
                raw = bytes(int(c) for c in vec[nr])
                L.ref_modem_push(ms[s][k], 1, raw, len(raw))
                L.ref_modem_push(ms[s][k], 2, raw[:-1], max(len(raw) - 1, 0))     # the shorter vector loses the `>=` comparison
                run(ms[s][k], s)
        elif not two and len(vec) == 1:
            raw = bytes(int(c) for c in vec[1])
            L.ref_modem_push(ms[s][0], 0, raw, len(raw))
            run(ms[s][0], s)

    for line in (tmp_path / "tap.txt").read_text().splitlines():
        parts = line.split(" ")
        kind, s = parts[0], int(parts[1])
        if kind == "D":
            flush()
            state["pending"] = (s, {})
            state["ret"] = 0
        elif kind == "R":
            # demodulate()'s return value: the reference's gr_modem::demodulate() on the same vectors -- false when getData() had
            # nothing (no vector was handed out), else synchronize()'s data_to_process; two-branch modes: either branch's synchroniser
            flush()
            returns["facade"].append(int(parts[2]))
            returns["reference"].append(state.get("ret", 0))
        elif kind == "B":
            state["pending"][1][int(parts[2])] = parts[3] if len(parts) > 3 else ""
        else:
            facade[s].append(" ".join(parts[2:]))
    flush()
    for per in ms:
        for m in per:
            L.ref_modem_free(m)
    for s in range(streams):
        assert facade[s] == reference[s], s
    assert sum(len(f) for f in facade) >= streams      # something was received (the loopback tests say what)
    assert returns["facade"] == returns["reference"] and 0 < sum(returns["facade"]) < len(returns["facade"])


def test_facade_spectrum_with_the_demodulator_valve_closed():
    """enable_demodulator(false) + enable_gui_fft(true): only the spectrum tap listens (gr_demod_base.cpp:1150-1153).  Twelve calls, each a tone
    at a different frequency per stream: the spectrum polled after a call must peak at THAT call's tone -- the upload, the FFT and the
    host's wait are on one stream (the FFT used to run on a stream nothing ordered behind the upload)."""
    r = subprocess.run([EXE, "valveoff", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("want") == 36


@pytest.mark.parametrize("mode,kind,fw,w,gain", [(9, "nbfm", 5000, 4000, 0), (14, "am", 5000, 4000, 0), (11, "usb", 2700, 2400, 500)])
def test_facade_rx_set_filter_width_and_set_gain(tmp_path, mode, kind, fw, w, gain):
    """gr_demod_base::set_filter_width(width, mode) / set_gain on the facade (called before the mode exists: its instance keeps them; a digital mode is
    ignored like the reference's default branch): the audio equals the oracle's chain with the reference setters' designs"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc
    import sig
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])
    if kind == "usb":
        n = 900000
        xs = [sig.make_ssb(n=n, seed=7, lsb=False), sig.make_ssb(n=n, seed=8, lsb=False)]
    else:
        n = 300000
        xs = [sig.make_analog(kind, n=n, seed=7)[0], sig.make_analog(kind, n=n, seed=8, gap=(40000, 240000))[0]]
    (tmp_path / "iq.bin").write_bytes(np.stack(xs).tobytes())
    r = subprocess.run([EXE, "analog", str(mode), "2", str(tmp_path / "iq.bin"), str(tmp_path / "a"), str(w), str(gain)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for s in range(2):
        got = np.fromfile(tmp_path / ("a%d.bin" % s), np.float32) + np.float32(0)
        if kind == "usb":
            want = orc.demod_ssb(xs[s], sb=0, filter_width=fw, set_width=w, gain=gain / 1000.0)["audio"] + np.float32(0)
        else:
            want = orc.demod_analog(xs[s], kind, filter_width=fw, set_width=w)["audio"] + np.float32(0)
        assert got.size == want.size // 640 * 640 and got.size >= 640
        assert np.array_equal(got.view(np.uint32), want[:got.size].view(np.uint32))


@pytest.mark.parametrize("mode,kind,fw,w,tone,rate,offset", [(9, "nbfm", 5000, 0, 0.0, 1000000, 0.0), (9, "nbfm", 5000, 4000, 88.5, 1000000, 0.0),
                                                             (14, "am", 5000, 4000, 0.0, 1000000, 0.0), (12, "lsb", 2700, 2400, 0.0, 1000000, 0.0),
                                                             (8, "nbfm", 2500, 0, 0.0, 2000000, 12500.0)])
def test_facade_tx_analog_set_audio(tmp_path, mode, kind, fw, w, tone, rate, offset):
    """gr_mod_base::set_audio / set_ctcss / set_filter_width on the TX facade (src/gr/gr_mod_base.cpp:793-797, 872-905): audio queued in ragged pieces on two
    radios, work() until the queues are empty; the IQ equals the oracle's modulator with the same setters applied (+ the back end's rotator and interpolator at
    a 2 Msps device rate), and the setters survive a mode change"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])
    n = 6 * 1024 + 700
    t = np.arange(n) / 8000.0
    audio = np.stack([0.6 * np.sin(2 * np.pi * 700 * t) + 0.3 * np.sin(2 * np.pi * 1500 * t), np.random.default_rng(51).uniform(-0.8, 0.8, n)]).astype(np.float32)
    (tmp_path / "audio.bin").write_bytes(audio.tobytes())
    r = subprocess.run([EXE, "analogtx", str(mode), "2", str(tmp_path / "audio.bin"), str(tmp_path / "iq"), str(w), str(tone), str(rate), str(offset)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for s in range(2):
        got = np.fromfile(tmp_path / ("iq%d.bin" % s), np.complex64)
        a = audio[s]
        if kind == "nbfm":
            a = a[:n // 4 * 4]                                   # the last n % 4 samples wait in the queue for a fourth one
            want = orc.mod_nbfm(a, filter_width=fw, bb_gain=0.75, set_width=w, ctcss=tone)
        elif kind == "am":
            want = orc.mod_am(a, filter_width=fw, bb_gain=0.75, set_width=w)
        else:
            want = orc.mod_ssb(a, sb=1, filter_width=fw, bb_gain=0.75, set_width=w)
        if rate != 1000000 or offset != 0.0:
            want = orc.tx_interp(orc.rotator(want, orc.phase_inc_to_turn(2 * np.pi * offset / 1000000.0)), rate)
        assert got.size == want.size and got.size > 0
        assert np.array_equal((got.view(np.float32) + np.float32(0)).view(np.uint32), (want.view(np.float32) + np.float32(0)).view(np.uint32))


def test_facade_tx_cw_key(tmp_path):
    """gr_mod_base::set_cw_k on the TX facade in CW600USB mode (src/gr/gr_mod_base.cpp:144,180,679-683,948-956): 3 calls key up, 4 down, 3 up of 1024 tone
    samples each -- the IQ equals the oracle's SSB chain over the keyed tone source"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])
    r = subprocess.run([EXE, "cw", "2", str(tmp_path / "iq")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    tone = np.concatenate([orc.sig_source_sin(8000, 600, 0.001, 3072, k0=0, offset=1.0), orc.sig_source_sin(8000, 600, 0.98, 4096, k0=3072, offset=1.0),
                           orc.sig_source_sin(8000, 600, 0.001, 3072, k0=7168, offset=1.0)])
    want = orc.mod_ssb(tone, sb=0, filter_width=1000)
    for s in range(2):
        got = np.fromfile(tmp_path / ("iq%d.bin" % s), np.complex64)
        assert got.size == want.size == 125 * 9 * 1024
        assert np.array_equal((got.view(np.float32) + np.float32(0)).view(np.uint32), (want.view(np.float32) + np.float32(0)).view(np.uint32))


@pytest.mark.parametrize("sr,fw", [(50000, 0.0), (0, 30000.0), (200000, 40000.0)])
def test_facade_scope_rate_and_filter_width(tmp_path, sr, fw):
    """gr_demod_base::set_time_sink_samp_rate / set_time_domain_filter_width on the facade (src/gr/gr_demod_base.cpp:1249-1301): what get_sample_data hands out
    equals the oracle's decimator with the reference's design for that setter sequence (a rate above 1 Msps is ignored)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])
    rng = np.random.default_rng(61)
    n = 400000
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.05 + 0.2 * np.exp(2j * np.pi * 3000.0 * np.arange(n) / 1e6)).astype(np.complex64)
    (tmp_path / "iq.bin").write_bytes(x.tobytes())
    r = subprocess.run([EXE, "scoperate", str(tmp_path / "iq.bin"), str(tmp_path / "s.bin"), str(sr), str(fw)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "s.bin", np.complex64)
    D = 1000000 // sr if sr else 10
    taps = orc.low_pass(1, 1000000, fw, fw) if fw > 0 else orc.low_pass(1, 1000000, sr // 2 - sr // 8, sr // 4)
    want = orc.decim_auto(orc.frontend(x, 1000000, 0.0), taps, D)
    # the sink hands out even counts; at most one item stays behind
    assert want.size - 1 <= got.size <= want.size and got.size > 1000
    assert np.array_equal((got.view(np.float32) + np.float32(0)).view(np.uint32), (want[:got.size].view(np.float32) + np.float32(0)).view(np.uint32))


def test_facade_tx_setDMRData(tmp_path):
    """gr_mod_base::setDMRData on the TX facade (src/gr/gr_mod_base.cpp:788-791 -> gr_dmr_source.cpp:56-73,100-127): every 33-byte frame is followed by 39 zero
    bytes with a "zero_samples" tag of 780 items on the first of them; two radios with different numbers of frames per call (the shorter queue is padded with
    zero bytes).  The IQ equals the oracle's gr_mod_dmr over the same bytes and tags (20 items of the zero-idle block's input per byte)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])
    frames = np.random.default_rng(71).integers(0, 256, (5, 33), dtype=np.uint8)
    (tmp_path / "frames.bin").write_bytes(frames.tobytes())
    r = subprocess.run([EXE, "dmrtx", str(tmp_path / "frames.bin"), str(tmp_path / "iq")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    z = np.zeros(39, np.uint8)
    # first run: stream 0 sends 144 bytes (two frames), stream 1 one frame + 72 bytes of padding; second run: one frame each
    s0 = np.concatenate([frames[0], z, frames[1], z, frames[2], z])
    s1 = np.concatenate([frames[3], z, np.zeros(72, np.uint8), frames[4], z])
    tags0 = [(20 * 33, 780), (20 * (72 + 33), 780), (20 * (144 + 33), 780)]
    tags1 = [(20 * 33, 780), (20 * (144 + 33), 780)]
    for s, (data, tags) in enumerate(((s0, tags0), (s1, tags1))):
        got = np.fromfile(tmp_path / ("iq%d.bin" % s), np.complex64)
        want = orc.mod_dmr(data, zero_runs=tags)
        assert got.size == want.size == 216 // 3 * 2500
        assert np.array_equal((got.view(np.float32) + np.float32(0)).view(np.uint32), (want.view(np.float32) + np.float32(0)).view(np.uint32)), "stream %d" % s
    # the tagged stretches are silence: the 39 zero bytes behind a frame (delayed by the block's 1439-item history), not the frame itself
    w = orc.mod_dmr(s0, zero_runs=tags0)
    lo, hi = (20 * 33 + 1439 - 62 + 60) * 125 // 3, (20 * 33 + 1439 - 62 + 780 - 60) * 125 // 3
    assert np.abs(w[lo:hi]).max() < 1e-3


def test_facade_rejected_settings_keep_the_handle_and_tx_setters_serialise(tmp_path):
    """(1) set_carrier_offset on a TX mode without the back end (M17, DSSS) throws and the modulator keeps running, a rejected set_mode keeps
    the previous modulator, the re-open of a mode with the back end drops what was queued for the old handle; (2) setDMRData from a second thread while
    work() runs: every frame's idle zeros are applied (the IQ of 24 frames equals the oracle's gr_mod_dmr with all 24 tags); (3) scope settings the engine
    rejects restore the previous ones and the demodulator keeps running; (4) a facade built for fewer than 1024 items per call keys CW."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), "adaptor"])
    r = subprocess.run([EXE, "robust", str(tmp_path / "dmr.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "robust ok" in r.stdout, r.stdout + r.stderr
    frames = np.array([[(0x1B + 7 * i + f) & 255 for i in range(33)] for f in range(24)], np.uint8)
    data = np.concatenate([np.concatenate([fr, np.zeros(39, np.uint8)]) for fr in frames])
    tags = [(20 * (72 * f + 33), 780) for f in range(24)]
    got = np.fromfile(tmp_path / "dmr.bin", np.complex64)
    want = orc.mod_dmr(data, zero_runs=tags)
    assert got.size == want.size
    assert np.array_equal((got.view(np.float32) + np.float32(0)).view(np.uint32), (want.view(np.float32) + np.float32(0)).view(np.uint32))
