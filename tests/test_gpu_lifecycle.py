"""Handle life cycle through the C ABI on the GPU: reset == fresh handle (the reference flushes the flowgraph on set_mode,
gr_demod_base.cpp:302-311, gr_mod_base.cpp:354-360), oversize calls are refused with QRL_ERR_TOO_BIG and leave the state intact,
unsupported configurations are refused at creation."""
import numpy as np
import pytest

import orc
import sig

pytestmark = pytest.mark.gpu


def test_demod_reset_equals_fresh_handle(qrl_ctx):
    import torch
    import qradiolink_amd as q
    iq = sig.make_batch("gmsk10k", 2, nframes=2, device_rate=1000000, seed=8)
    n = iq.shape[1] & ~1
    d = torch.from_numpy(iq[:, :n]).cuda()
    dem = q.Demod(qrl_ctx, q.MODEM_GMSK10K, batch=2, max_chunk=n)
    first = q.collect(dem, d, n)
    dem.process_async(d[:, : n // 2 & ~1])          # leave the handle in the middle of a stream
    dem.sync()
    dem.reset()
    again = q.collect(dem, d, n)
    dem.close()
    for port in ("bits_a", "bits_b", "filtered", "constellation"):
        for b in range(2):
            assert np.array_equal(np.asarray(first[port][b]).view(np.uint8), np.asarray(again[port][b]).view(np.uint8)), port


def test_oversize_call_is_refused_and_state_survives(qrl_ctx):
    import torch
    import qradiolink_amd as q
    iq = sig.make_batch("2fsk1k", 1, nframes=2, device_rate=1000000, seed=9)
    n = iq.shape[1] & ~1
    d = torch.from_numpy(iq[:, :n]).cuda()
    dem = q.Demod(qrl_ctx, q.MODEM_2FSK1K, batch=1, max_chunk=n // 2 + 2)
    half = (n // 2) & ~1
    parts = []
    dem.process_async(d[:, :half]); dem.sync()
    c = dem.counts.cpu().numpy(); parts.append(dem.bits_a.cpu().numpy()[0, :c[0, 2]].copy())
    with pytest.raises(q.QrlError) as e:
        dem.process_async(d)                      # n > max_chunk
    assert "max_chunk" in str(e.value)
    dem.process_async(d[:, half:2 * half]); dem.sync()
    c = dem.counts.cpu().numpy(); parts.append(dem.bits_a.cpu().numpy()[0, :c[0, 2]].copy())
    dem.close()
    ref = orc.demod_2fsk(orc.frontend(iq[0, :2 * half], 1000000, 0.0))
    assert np.array_equal(np.concatenate(parts), ref["bits_a"])


def test_unsupported_configurations_are_refused(qrl_ctx):
    import qradiolink_amd as q
    for modem in (13, 28, 29, 38):              # CW600USB, FREEDV1600USB, FREEDV700CUSB, MMDVM: not on this path
        with pytest.raises(q.QrlError):
            q.Demod(qrl_ctx, modem, batch=1, max_chunk=1024)
    with pytest.raises(q.QrlError):
        q.Demod(qrl_ctx, q.MODEM_GMSK10K, batch=1, max_chunk=1024, device_samp_rate=1500000)   # not 1e6 / multiple of 1e6 >= 2e6
    with pytest.raises(q.QrlError):
        q.Mod(qrl_ctx, 12, batch=1, max_bytes=16)                                                 # LSB2500
    with pytest.raises(q.QrlError):
        q.Channelizer(qrl_ctx, 65, batch=1, max_chunk=65 * 100)
    with pytest.raises(q.QrlError):
        q.Synth(qrl_ctx, 8, batch=1, max_samples=100)                                             # > MAX_MMDVM_CHANNELS
    with pytest.raises(q.QrlError):
        q.Deframer(qrl_ctx, 4, 1)


def test_mod_and_channelizer_reset(qrl_ctx):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(1)
    data = torch.from_numpy(rng.integers(0, 256, (2, 64), dtype=np.uint8)).cuda()
    mod = q.Mod(qrl_ctx, q.MODEM_GMSK10K, batch=2, max_bytes=64)
    a = mod.process(data).cpu().numpy()
    mod.process(data)
    mod.reset()
    b = mod.process(data).cpu().numpy()
    mod.close()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    x = (0.1 * (rng.standard_normal((1, 4000)) + 1j * rng.standard_normal((1, 4000)))).astype(np.complex64)
    ch = q.Channelizer(qrl_ctx, 10, batch=1, max_chunk=4000)
    dx = torch.from_numpy(x).cuda()
    o1, c1 = ch.process(dx); o1, c1 = o1.cpu().numpy().copy(), c1.cpu().numpy().copy()
    ch.process(dx)
    assert ch.lib.qrl_chan_reset(ch.h) == 0
    o2, c2 = ch.process(dx); o2, c2 = o2.cpu().numpy(), c2.cpu().numpy()
    ch.close()
    assert np.array_equal(c1, c2) and all(np.array_equal(o1[0, k, :c1[0, k]], o2[0, k, :c2[0, k]]) for k in range(10))


def test_process_host_matches_device_path(qrl_ctx, capsys):
    """qrl_demod_process_host (H2D copy + one pass + D2H of the bit ports, what a host-buffer caller uses) returns the same
    bits as the device-pointer path; prints the PCIe-inclusive rate of the call for DESIGN.md"""
    import ctypes as C
    import time
    import torch
    import qradiolink_amd as q
    B = 32
    iq = sig.make_batch("gmsk10k", 2, nframes=2, device_rate=4000000, rx_offset_hz=25000.0, seed=12)
    n = iq.shape[1] & ~1
    iq = np.ascontiguousarray(np.concatenate([iq[:, :n]] * (B // 2)))
    dem = q.Demod(qrl_ctx, q.MODEM_GMSK10K, batch=B, max_chunk=n, device_samp_rate=4000000, carrier_offset_hz=25000.0)
    out = q.collect(dem, torch.from_numpy(iq).cuda(), n)
    dem.reset()
    fcap, ccap, bcap = C.c_size_t(), C.c_size_t(), C.c_size_t()
    assert dem.lib.qrl_demod_out_caps(dem.h, n, C.byref(fcap), C.byref(ccap), C.byref(bcap)) == 0
    ba = np.zeros((B, bcap.value), np.uint8); bb = np.zeros((B, bcap.value), np.uint8); cnt = np.zeros((B, 4), np.uint32)
    t0 = time.perf_counter()
    rc = dem.lib.qrl_demod_process_host(dem.h, iq.ctypes.data_as(C.c_void_p), n, n, ba.ctypes.data_as(C.c_void_p),
                                        bb.ctypes.data_as(C.c_void_p), bcap.value, cnt.ctypes.data_as(C.c_void_p))
    dt = time.perf_counter() - t0
    assert rc == 0
    dem.close()
    for b in range(B):
        assert np.array_equal(ba[b, :cnt[b, 2]], out["bits_a"][b]) and np.array_equal(bb[b, :cnt[b, 3]], out["bits_b"][b])
    with capsys.disabled():
        print("\n[process_host] %d x %d samples from pageable host memory: %.1f ms = %.2f GS/s (%.1f GB/s over PCIe incl. allocation)"
              % (B, n, dt * 1e3, B * n / dt / 1e9, B * n * 8 / dt / 1e9))
