"""PINS the L1 frame synchroniser (oracle orc_modem_sync = the contract of k_framesync) and the mode
tables against the REAL reference: oracle/_ref/libqrl_ref.so holds src/gr_modem.cpp itself (class gr_modem: toggleRxMode, demodulate,
synchronize, findSync, packBytes, processReceivedData), compiled unmodified against oracle/qt_stub (a sliver of Qt; the signals moc
would generate are defined by oracle/ref_shim_modem.cpp and record their arguments; gr_demod_base is a mailbox the test fills).
The same bit stream goes through the reference class (ragged getData() vectors) and through the oracle; the reference's signals must
equal what processReceivedData (src/gr_modem.cpp:1285-1441) makes of the oracle's (frame type, payload) records."""
import ctypes as C
import os

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libqrl_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libqrl_ref.so not built (make -C oracle ref needs /root/reference)")

FT = dict(Voice2=0xED89, Voice1=0xB5, Text=0x89EDAA, IP=0xDE98AA, Video=0x98DEAA, Callsign=0x8CC8DD, Proto=0xED77AA, End=0x4C8A2B)
TWO_BRANCH = (0, 15, 17, 19, 20, 21, 22, 24, 25, 16, 18)      # gr_modem.cpp:1043-1053


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(REF)
    vp = C.c_void_p
    L.ref_modem_new.restype = vp
    for name, args in dict(ref_modem_free=[vp], ref_modem_init_rx=[vp, C.c_int], ref_modem_init_tx=[vp, C.c_int], ref_modem_toggle_rx=[vp, C.c_int],
                           ref_modem_toggle_tx=[vp, C.c_int], ref_modem_rx_frame_length=[vp], ref_modem_tx_frame_length=[vp], ref_modem_bit_buf_len=[vp],
                           ref_modem_push=[vp, C.c_int, vp, C.c_size_t], ref_modem_demodulate=[vp], ref_modem_events=[vp, vp, C.c_size_t]).items():
        getattr(L, name).argtypes = args
    L.ref_modem_events.restype = C.c_size_t
    return L


def _events(ref, m):
    buf = C.create_string_buffer(1 << 22)
    n = ref.ref_modem_events(m, buf, len(buf))
    return buf.raw[:n].decode("latin-1").splitlines()


def test_mode_tables_equal_the_reference(ref):
    """toggleRxMode (gr_modem.cpp:203-322): bit buffer length and frame length of every modem type the oracle knows"""
    m = ref.ref_modem_new()
    ref.ref_modem_init_rx(m, 0)
    for mt in list(range(0, 8)) + list(range(15, 28)) + [40]:
        if mt == 23:
            continue
        ref.ref_modem_toggle_rx(m, mt)
        bl, fl = C.c_int(), C.c_int()
        orc.lib.orc_modem_sync_geometry(mt, C.byref(bl), C.byref(fl))
        assert (ref.ref_modem_bit_buf_len(m), ref.ref_modem_rx_frame_length(m)) == (bl.value, fl.value), mt
    ref.ref_modem_free(m)


def _expected(records_with_sync, rx_len):
    """processReceivedData over (frame_type, payload, _modem_sync at that moment) records"""
    ev, last = [], None
    for ft, p, msync in records_with_sync:
        if ft == FT["End"]:
            if last == FT["Text"]:
                ev.append("text 0a")
            ev += ["endaudio", "receiveend"]
        elif ft == FT["Text"]:
            ev.append("dataframe")
            last = ft
            ev.append("text " + p[:rx_len].rstrip(b"\x00").hex())
        elif ft == FT["Proto"]:
            ev += ["dataframe", "proto " + p[:rx_len].hex()]
            last = ft
        elif ft == FT["Callsign"]:
            last = ft
            s = p[:7].split(b"\x00")[0]
            s = bytes(c for c in s if chr(c).isascii() and (chr(c).isalnum() or c in b"/ \t\n\r\x0b\x0c"))[:7]
            ev.append("callsign " + s.decode("latin-1"))
        elif ft == FT["Voice1"]:
            last = ft
            if msync >= 16:
                ev.append("audio " + p[:rx_len].hex())
        elif ft == FT["Voice2"]:
            last = ft
            ev.append("audio " + p[1:1 + rx_len].hex())
        elif ft == FT["Video"]:
            ev += ["dataframe", "video " + p[:rx_len].hex()]
            last = ft
        elif ft == FT["IP"]:
            last = ft
            ev.append("net " + p[:rx_len].hex())
    return ev


def _stream(rng, mt, nframes):
    """random bits with frames of every type the mode's class knows in between (some sync words damaged, some cut short)"""
    bl, fl = C.c_int(), C.c_int()
    cls = orc.lib.orc_modem_sync_geometry(mt, C.byref(bl), C.byref(fl))
    kinds = {0: ["Voice1"], 1: ["IP", "Video", "End"], 2: ["Voice2", "Text", "Proto", "Callsign", "End", "Video"]}[cls]
    bits = [rng.integers(0, 2, 50)]
    for k in range(nframes):
        kind = kinds[int(rng.integers(0, len(kinds)))]
        w = FT[kind]
        word = np.unpackbits(np.array(list(w.to_bytes(3 if w > 0xFFFF else 2 if w > 0xFF else 1, "big")), np.uint8))
        if k % 7 == 6:
            word = word.copy(); word[int(rng.integers(0, word.size))] ^= 1
        n = bl.value + 8
        payload = rng.integers(0, 2, n)
        if kind in ("Text", "Callsign"):        # printable text with a zero tail
            txt = bytes(rng.integers(0x20, 0x7F, int(rng.integers(1, fl.value + 1)), dtype=np.uint8)) + bytes(fl.value + 2)
            payload = np.unpackbits(np.frombuffer(txt, np.uint8))[:n]
        bits += [word, payload[: n if k % 5 else n // 2], rng.integers(0, 2, int(rng.integers(0, 40)))]
        if cls == 0 and k % 3:                  # 1k modes: back-to-back frames open the _modem_sync >= 16 gate
            bits.pop()
    return np.concatenate(bits).astype(np.uint8), fl.value


@pytest.mark.parametrize("mt", [26, 27, 22, 19, 7, 3, 18, 21, 0, 24])
def test_frame_synchroniser_oracle_equals_the_reference_class(ref, mt):
    rng = np.random.default_rng(100 + mt)
    bits, rx_len = _stream(rng, mt, 60 if mt not in (26, 27) else 12)
    # ---- the reference: ragged vectors through gr_demod_base::getData; two-branch modes get the stream on branch 1 and a shorter
    # vector of noise on branch 2 (the `>=` rule then always takes branch 1, gr_modem.cpp:1080-1090)
    m = ref.ref_modem_new()
    ref.ref_modem_init_rx(m, mt)
    got, pos = [], 0
    while pos < bits.size:
        n = int(rng.integers(32, 3000))
        part = np.ascontiguousarray(bits[pos:pos + n])
        if mt in TWO_BRANCH:
            ref.ref_modem_push(m, 1, part.ctypes.data, part.size)
            other = np.ascontiguousarray(rng.integers(0, 2, max(part.size - 1, 1), dtype=np.uint8))
            ref.ref_modem_push(m, 2, other.ctypes.data, other.size)
        else:
            ref.ref_modem_push(m, 0, part.ctypes.data, part.size)
        ref.ref_modem_demodulate(m)
        got += _events(ref, m)
        pos += n
    ref.ref_modem_free(m)
    # ---- the oracle, bit by bit so that _modem_sync is known at the moment a frame completes
    ms = orc.ModemSync(mt)
    recs = []
    for b in bits:
        for ft, p in ms.feed(np.array([b], np.uint8)):
            recs.append((ft, p, int(ms.st[4])))
    assert len(recs) >= 8
    assert got == _expected(recs, rx_len)


def test_m17_frames_through_the_reference_gr_modem(ref):
    """mode 40: the oracle's frame synchroniser (class 3: LSF 0x55F7, stream 0xFF5D, EOT 0x555D555D; 46-byte frames) and frame decoder
    against the reference's gr_modem, which runs its own M17FrameDecoder on what its synchroniser cuts out: frames made by the
    reference's M17FrameEncoder, embedded in noise bits.  Every stream frame's 16 payload bytes must come out as digitalAudio, in
    order, and equal what orc_m17_decode_frame makes of the oracle's records; the LSF produces one m17FrameInfoReceived."""
    import test_framefec as F
    L = C.CDLL(REF)
    rng = np.random.default_rng(40)
    lsf28 = rng.integers(0, 256, 28, dtype=np.uint8)
    lsf28[:12] = [0, 0, 0x4B, 0x13, 0xD1, 0x06, 0, 0, 0x4B, 0x13, 0xD1, 0x07]      # two ordinary base-40 call signs
    pl = rng.integers(0, 256, (12, 16), dtype=np.uint8)
    frames = np.zeros((13, 48), np.uint8)
    L.ref_m17_encode(F.P(lsf28), F.P(pl), 12, F.P(frames))
    bits = [rng.integers(0, 2, 100, dtype=np.uint8)]
    for f in frames:
        bits += [np.unpackbits(f), rng.integers(0, 2, int(rng.integers(0, 30)), dtype=np.uint8)]
    bits = np.concatenate(bits)
    m = ref.ref_modem_new()
    ref.ref_modem_init_rx(m, 40)
    got, pos = [], 0
    while pos < bits.size:
        n = int(rng.integers(32, 700))
        part = np.ascontiguousarray(bits[pos:pos + n])
        ref.ref_modem_push(m, 0, part.ctypes.data, part.size)
        ref.ref_modem_demodulate(m)
        got += _events(ref, m)
        pos += n
    ref.ref_modem_free(m)
    ms = orc.ModemSync(40)
    recs = ms.feed(bits)
    audio = []
    for ft, p in recs:
        frame = np.frombuffer(ft.to_bytes(2, "big") + p[:46], np.uint8) if ft in (0x55F7, 0xFF5D) else None
        if frame is not None:
            rec = F.orc_m17_records(frame[None, :])[0]
            if rec[0] == 2:
                audio.append("audio " + bytes(rec[4:20]).hex())
    assert [ft for ft, _ in recs].count(0xFF5D) >= 12 and [ft for ft, _ in recs].count(0x55F7) >= 1
    assert [e for e in got if e.startswith("audio ")] == audio and len(audio) >= 12
    assert audio[:12] == ["audio " + bytes(p).hex() for p in pl]
    assert sum(e.startswith("m17info ") for e in got) == 1


@pytest.mark.parametrize("mt", [26, 22, 3, 18, 24])
def test_demodulate_return_value_equals_the_reference_class(ref, mt):
    """gr_modem::demodulate() returns synchronize()'s data_to_process: true iff a bit of the call was collected while a sync was held (src/gr_modem.cpp:1121-1175),
    also while a frame is only partly there.  The oracle's orc_modem_sync_collected() (what qrl_framesync_set_activity_output exports from k_framesync,
    tests/test_gpu_deframe.py) must say the same, call by call, for ragged calls over a stream with damaged sync words and frames cut short."""
    rng = np.random.default_rng(300 + mt)
    bits, _ = _stream(rng, mt, 40 if mt not in (26, 27) else 10)
    m = ref.ref_modem_new()
    ref.ref_modem_init_rx(m, mt)
    ms = orc.ModemSync(mt)
    pos, calls, trues = 0, 0, 0
    while pos < bits.size:
        n = int(rng.integers(32, max(40, bits.size // 60)))
        part = np.ascontiguousarray(bits[pos:pos + n])
        if part.size < 32:
            break   # gr_bit_sink::get_data hands out nothing below 32 bits (src/gr/gr_bit_sink.cpp:45-59): demodulate() returns false without consuming
        if mt in TWO_BRANCH:
            ref.ref_modem_push(m, 1, part.ctypes.data, part.size)
            other = np.ascontiguousarray(rng.integers(0, 2, max(part.size - 1, 1), dtype=np.uint8))
            ref.ref_modem_push(m, 2, other.ctypes.data, other.size)
        else:
            ref.ref_modem_push(m, 0, part.ctypes.data, part.size)
        want = bool(ref.ref_modem_demodulate(m))
        _events(ref, m)
        ms.feed_raw(part)
        assert (ms.collected > 0) == want, (calls, pos, n, ms.collected, want)
        calls += 1; trues += want
        pos += n
    ref.ref_modem_free(m)
    assert calls > 20 and 0 < trues < calls
