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
