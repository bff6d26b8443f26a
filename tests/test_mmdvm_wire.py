"""a35 / a50 host layer (qradiolink_amd/host/mmdvm_wire.cpp through the ctypes shim tests/host/libmmdvm_shim.so) against the
Python model tests/mmdvm_model.py and against known answers read off the reference: the 720-sample wire frame
{u32 n, u32 rssi, u8 control[n], i16 data[n]} (gr_mmdvm_sink.cpp:155-165), slot marks 0x08 / 0x04, the idle-slot zero tag of
750 samples and the tx_time tag at item 710 (gr_mmdvm_source.cpp:23,112-128)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import mmdvm_model as M

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHIM = os.path.join(HERE, "host", "libmmdvm_shim.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(SHIM):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qradiolink_amd", "csrc"), SHIM.replace(ROOT, "../..")])
    L = C.CDLL(SHIM)
    vp = C.c_void_p
    L.mw_timer_new.restype = vp
    L.mw_timer_free.argtypes = [vp]
    L.mw_timer_set_timer.argtypes = [vp, C.c_uint64, C.c_int]
    L.mw_timer_set_params.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mw_timer_allocate_slot.restype = C.c_uint64
    L.mw_timer_allocate_slot.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    L.mw_timer_pending.restype = C.c_uint64
    L.mw_timer_pending.argtypes = [vp, C.c_int]
    L.mw_sink_new.restype = vp
    L.mw_sink_new.argtypes = [vp, C.c_int, C.c_int]
    L.mw_sink_free.argtypes = [vp]
    L.mw_sink_work.argtypes = [vp, C.c_int, C.c_int] + [vp] * 7
    L.mw_sink_take.restype = C.c_size_t
    L.mw_sink_take.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_int)]
    L.mw_source_new.restype = vp
    L.mw_source_new.argtypes = [vp, C.c_int, C.c_int]
    L.mw_source_free.argtypes = [vp]
    L.mw_source_push.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t]
    L.mw_source_work.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.mw_zero_runs.argtypes = [vp, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, vp, C.c_int]
    return L


def _sink_call(lib, sink, samples, rssi, tags):
    nch, n = len(samples), len(samples[0])
    a = np.ascontiguousarray(np.array(samples, np.int16))
    r = np.array([v for ch in rssi for v in ch], np.float32)
    nr = np.array([len(ch) for ch in rssi], np.int32)
    flat = [t for ch in tags for t in ch]
    off = np.array([t[0] for t in flat], np.uint32)
    secs = np.array([t[1] for t in flat], np.uint64)
    fr = np.array([t[2] for t in flat], np.float64)
    nt = np.array([len(ch) for ch in tags], np.int32)
    p = lambda x: x.ctypes.data if x.size else None
    assert lib.mw_sink_work(sink, nch, n, a.ctypes.data, p(r), nr.ctypes.data, p(off), p(secs), p(fr), nt.ctypes.data) == n
    buf = np.zeros(1 << 16, np.uint8)
    nf = C.c_int()
    nb = lib.mw_sink_take(sink, buf.ctypes.data, buf.size, C.byref(nf))
    out, at = [], 0
    while at < nb:
        ch, ln = struct.unpack_from("<II", buf, at)
        out.append((ch, bytes(buf[at + 8:at + 8 + ln])))
        at += 8 + ln
    assert len(out) == nf.value
    return out


def test_sink_frames_and_slot_marks(lib):
    """three channels, ragged call sizes, the time base set by an rx_time tag, slots allocated by the TX side in between"""
    rng = np.random.default_rng(1)
    nch = 3
    tm, sink = lib.mw_timer_new(), None
    sink = lib.mw_sink_new(tm, nch, 1)
    mt = M.Timer()
    ms = M.Sink(mt, nch)
    lib.mw_timer_set_params(tm, 720, 41667, 30000000, 20000000)   # 20 ms burst delay (the default constructor's 1e14 ns is replaced
    mt.set_params(20000000)                                       #  by the radio controller before the graph runs)
    frames_c, frames_m = [], []
    t0 = 5 * 1000000000 + 250000000
    pos = 0
    for call, n in enumerate([720, 300, 720, 411, 720, 720, 9, 720, 720, 720, 720, 600]):
        samples = [list(rng.integers(-3000, 3000, n, dtype=np.int16)) for _ in range(nch)]
        rssi = [[float(-60 - ch - 0.1 * k) for k in range((pos + n) // 300 - pos // 300)] for ch in range(nch)]
        tags = [[(0, 5, 0.25)] if call == 0 else [] for _ in range(nch)]
        if call in (1, 3, 4, 8):   # the TX side (gr_mmdvm_source) allocates slots from its own thread: same calls on both timers
            for ch in range(nch):
                tc = C.c_int64(-1)
                assert lib.mw_timer_allocate_slot(tm, 1 + (call + ch) % 2, ch, C.byref(tc)) == mt.allocate_slot(1 + (call + ch) % 2, ch)[0]
        frames_c += _sink_call(lib, sink, samples, rssi, tags)
        frames_m += ms.work([[int(v) for v in s] for s in samples], rssi, tags)
        pos += n
    assert frames_c == frames_m and len(frames_c) >= 8 * nch
    # known answers: header, layout, mark values
    ch, msg = frames_c[0]
    n, rssi0 = struct.unpack_from("<II", msg, 0)
    assert n == 720 and len(msg) == 8 + 720 + 1440
    marks = [(c, i, m[8 + i]) for c, m in frames_c for i in range(720) if m[8 + i]]
    assert marks and all(v in (0x08, 0x04) for _, _, v in marks)
    assert any(struct.unpack_from("<II", m, 0)[1] > 0 for _, m in frames_c)      # a slot's RSSI = lower of its last two tags
    lib.mw_sink_free(sink)
    lib.mw_timer_free(tm)
    assert t0 > 0


def test_slot_mark_position_known_answer(lib):
    """time base 7.0 s at item 0; a slot allocated at once is due BURST_DELAY later: 1e14 ns / 41667 ns per sample items on"""
    tm = lib.mw_timer_new()
    sink = lib.mw_sink_new(tm, 1, 1)
    lib.mw_timer_set_timer(tm, 7 * 10 ** 9, 0)
    tc = C.c_int64(0)
    nsec = lib.mw_timer_allocate_slot(tm, 2, 0, C.byref(tc))
    assert nsec == 7 * 10 ** 9 + M.BURST_DELAY     # _last_slot = elapsed (first allocation), + burst delay (bursttimer.cpp:240-280)
    lib.mw_sink_free(sink)
    lib.mw_timer_free(tm)


def test_source_bursts_tags_and_idle_slots(lib):
    rng = np.random.default_rng(2)
    nch = 2
    tm = lib.mw_timer_new()
    src = lib.mw_source_new(tm, nch, 1)
    mt = M.Timer()
    ms = M.Source(mt, nch, True)
    out = np.zeros((nch, 720), np.int16)
    tags = np.zeros(4 * 64, np.uint64)
    nt, sl = C.c_int(), C.c_int64()
    # before the RX side has set the time base the TDMA source produces nothing (gr_mmdvm_source.cpp:190-204)
    assert lib.mw_source_work(src, nch, out.ctypes.data, tags.ctypes.data, 64, C.byref(nt), C.byref(sl)) == 0
    assert ms.work([b"", b""])[0] == 0
    for ch in range(nch):
        lib.mw_timer_set_timer(tm, 3 * 10 ** 9, ch)
        mt.set_timer(3 * 10 ** 9, ch)
    for step in range(10):
        msgs = []
        for ch in range(nch):
            if (step + ch) % 3 == 0:
                msgs.append(b"")
                continue
            n = 720
            ctrl = bytearray(n)
            ctrl[0] = M.MARK_SLOT1 if (step + ch) % 2 else M.MARK_SLOT2
            data = rng.integers(-20000, 20000, n, dtype=np.int16)
            m = struct.pack("<I", n) + bytes(ctrl) + data.tobytes()
            msgs.append(m)
            lib.mw_source_push(src, ch, m, len(m))
        assert lib.mw_source_work(src, nch, out.ctypes.data, tags.ctypes.data, 64, C.byref(nt), C.byref(sl)) == 720
        items, mout, mtags, msleep = ms.work(msgs)
        got_tags = [tuple(int(v) for v in tags[4 * i:4 * i + 4]) for i in range(nt.value)]
        assert got_tags == [tuple(t) for t in mtags], step
        assert np.array_equal(out, np.array(mout, np.int16)), step
        assert sl.value == msleep
        for ch in range(nch):
            if msgs[ch] == b"":      # idle slot: zeros, a zero_samples tag of 750 at item 0, (channel 0) a tx_time tag at item 710
                assert not out[ch].any() and (ch, 0, 1, 750) in got_tags
                if ch == 0:
                    assert any(t[0] == 0 and t[1] == 710 and t[2] == 0 for t in got_tags)
    lib.mw_source_free(src)
    lib.mw_timer_free(tm)


def test_zero_idle_runs_scale_with_the_resampler(lib):
    tags = [(0, 0, 1, 750), (1, 0, 1, 750), (0, 5, 0, 123456), (0, 700, 1, 750), (0, 719, 1, 750)]
    t = np.array(tags, np.uint64)
    runs = np.zeros(32, np.uint64)
    for written, num, den in ((0, 25, 24), (720 * 7, 25, 24), (720 * 3 + 11, 1, 1)):
        n = lib.mw_zero_runs(t.ctypes.data, len(tags), 0, written, num, den, runs.ctypes.data, 16)
        got = [(int(runs[2 * i]), int(runs[2 * i + 1])) for i in range(n)]
        assert got == M.zero_runs(tags, 0, written, num, den)
    # an idle slot starts on a slot boundary: 720 k items at 24 ksps = exactly 750 k at 25 ksps, the run covers the whole slot
    assert M.zero_runs([(0, 0, 1, 750)], 0, 720 * 4, 25, 24) == [(3000, 750)]
