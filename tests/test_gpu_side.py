"""GPU parity of the side outputs of gr_demod_base: rssi_block on demodulator port 0 (reference src/gr/rssi_block.cpp:31-44,
gr_demod_base.cpp:199-200) bit for bit against the oracle over ragged calls, and rx_fft_c (src/gr/rx_fft.cpp:71-131): its
fill / transform / hold state machine and the power spectrum against the oracle's float64 FFT."""
import numpy as np
import pytest

import orc
import sig

pytestmark = pytest.mark.gpu


def test_rssi_block_bit_exact_over_ragged_calls(qrl_ctx):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(11)
    B, total = 70, 9300                      # two workgroups (64 + 6 streams), windows and restarts at 2000, 4000, ...
    x = np.zeros((B, total), np.complex64)
    for b in range(B):
        amp = np.where(np.arange(total) < 3000 + 37 * b, 0.01 * (1 + b % 5), 0.2)
        x[b] = (amp * (rng.standard_normal(total) + 1j * rng.standard_normal(total))).astype(np.complex64)
    x[3, 100:2300] = 0                       # a stretch of exact zeros
    d = torch.from_numpy(x).cuda()
    r = q.Rssi(qrl_ctx, B, level=-20.0)
    got = [[] for _ in range(B)]
    cnts = np.zeros(B, np.int64)
    pos = 0
    for c in [1, 63, 64, 65, 1999, 1, 2000, 3107, 2000]:
        counts = np.array([max(c - (b % 4), 0) for b in range(B)], np.int32)    # ragged per-stream counts
        # stream b consumes from its own cursor: build the call's input rows
        rows = np.zeros((B, c), np.complex64)
        for b in range(B):
            rows[b, :counts[b]] = x[b, cnts[b]:cnts[b] + counts[b]]
        out, last = r.process(torch.from_numpy(rows).cuda(), counts=torch.from_numpy(counts).cuda(), count_stride=1, n=c)
        out, last = out.cpu().numpy(), last.cpu().numpy()
        for b in range(B):
            got[b].append(out[b, :counts[b]].copy())
            if counts[b]:
                assert last[b] == out[b, counts[b] - 1]
        cnts += counts
        pos += c
    r.close()
    for b in range(B):
        g = np.concatenate(got[b])
        w = orc.rssi_block(x[b, :cnts[b]], level=-20.0)
        assert g.size == w.size and np.array_equal(g.view(np.uint32), w.view(np.uint32)), "stream %d" % b
    g3 = np.concatenate(got[3])
    assert g3[2250] < g3[90] - 30.0          # 2200 zero samples: the window has emptied, the IIR has decayed
    del d


def test_rssi_on_demodulator_port0(qrl_ctx):
    """gr_demod_base wiring: qrl_demod_process port 0 (+ its device-side counts) -> qrl_rssi_process, against the oracle chain"""
    import torch
    import qradiolink_amd as q
    iq = sig.make_batch("gmsk10k", 3, nframes=3, device_rate=1000000, seed=5)
    dem = q.Demod(qrl_ctx, 22, batch=3, max_chunk=1 << 17)
    r = q.Rssi(qrl_ctx, 3, level=0.0)
    d = torch.from_numpy(iq).cuda()
    got = [[] for _ in range(3)]
    pos = 0
    while pos < iq.shape[1]:
        n = min(1 << 17, iq.shape[1] - pos) & ~1
        if not n:
            break
        o = dem.process(d[:, pos:pos + n].contiguous())
        c = o["counts"][:, 0].contiguous()
        out, _ = r.process(o["filtered"], counts=c, count_stride=1)
        cc = c.cpu().numpy()
        for b in range(3):
            got[b].append(out[b, :cc[b]].cpu().numpy().copy())
        pos += n
    for b in range(3):
        want = orc.rssi_block(orc.demod_gmsk(iq[b, :pos], sps=1, filter_width=20000)["filtered"], 0.0)
        g = np.concatenate(got[b])
        assert g.size == want.size and np.array_equal(g.view(np.uint32), want.view(np.uint32))
    dem.close()
    r.close()


@pytest.mark.parametrize("n,wintype", [(4096, 5), (32768, 5), (1024, 0), (2048, 4), (512, 7), (256, 6)])
def test_rx_fft_spectrum_and_state_machine(qrl_ctx, n, wintype):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(n + wintype)
    B = 3
    t = np.arange(3 * n + 10)
    x = np.stack([(0.3 * np.exp(2j * np.pi * (0.07 + 0.11 * b) * t) + 0.01 * (rng.standard_normal(t.size) + 1j * rng.standard_normal(t.size))).astype(np.complex64)
                  for b in range(B)])
    d = torch.from_numpy(x).cuda()
    f = q.Fft(qrl_ctx, B, fftsize=n, wintype=wintype)
    assert f.get_fft_size() == n and f.get_window_type() == wintype
    f.work(d[:, :n])
    assert f.get_fft_data() is None                       # a new block is disabled (rx_fft.cpp:58): nothing was taken
    f.set_enabled(True)
    f.work(d[:, :n // 2].contiguous())
    f.work(d[:, n // 2:n].contiguous())
    assert f.get_fft_data() is None                       # buffer full, but the transform runs when the NEXT sample arrives (:85-92)
    f.work(d[:, n:n + 7].contiguous())                    # -> FFT of samples [0, n), then 7 samples go into the new buffer
    f.work(d[:, n + 7:2 * n].contiguous())                # d_push > 0: ignored until somebody reads
    got = f.get_fft_data()
    assert got is not None and got.shape == (B, n)
    got = got.cpu().numpy()
    assert f.get_fft_data() is None                       # data_ready was cleared
    win = {0: np.hamming, 1: np.hanning, 2: np.blackman, 6: np.bartlett}.get(wintype)
    if win is not None:
        w = win(n).astype(np.float32)
    elif wintype == 5:
        k = 2 * np.pi * np.arange(n) / (n - 1)
        w = (0.35875 - 0.48829 * np.cos(k) + 0.14128 * np.cos(2 * k) - 0.01168 * np.cos(3 * k)).astype(np.float32)
    elif wintype == 4:
        w = np.kaiser(n, 6.76).astype(np.float32)
    else:
        k = 2 * np.pi * np.arange(n) / (n - 1)
        w = ((1.0 - 1.93 * np.cos(k) + 1.29 * np.cos(2 * k) - 0.388 * np.cos(3 * k) + 0.0322 * np.cos(4 * k)) / 4.6402).astype(np.float32)
    for b in range(B):
        want = orc.power_spectrum(x[b, :n], w)
        strong = want > want.max() - 80.0                 # float32 FFT against float64: compare where the spectrum is not rounding noise
        assert np.max(np.abs(got[b][strong] - want[strong])) < 0.05
        assert np.argmax(got[b]) == np.argmax(want)
    # the second frame: 7 samples taken before the hold, the rest after the read
    f.work(d[:, 2 * n:3 * n - 7].contiguous())
    f.work(d[:, 3 * n - 7:3 * n + 1].contiguous())
    got2 = f.get_fft_data().cpu().numpy()
    frame = np.concatenate([x[:, n:n + 7], x[:, 2 * n:3 * n - 7]], axis=1)
    for b in range(B):
        want = orc.power_spectrum(frame[b], w)
        strong = want > want.max() - 80.0
        assert np.max(np.abs(got2[b][strong] - want[strong])) < 0.05
    f.set_fft_size(n // 2)
    assert f.get_fft_size() == n // 2 and f.get_fft_data() is None
    f.close()
