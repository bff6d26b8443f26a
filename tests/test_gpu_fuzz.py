"""Seeded random call sizes (tiny calls of 2 .. 200 samples mixed with large ones) against the one-shot oracle, for every front-end
kernel and every tail family: the rings, the carried histories, the edge outputs of the phase-lane decimators (a call too short to
hold one interior output, a call that ends inside the warm-up of the next) and the pipelined streams must not depend on how the
stream is cut (ragged inputs)."""
import numpy as np
import pytest

import orc
import sig
from test_gpu_parity import _oracle

pytestmark = pytest.mark.gpu


def _random_cuts(rng, total, rate):
    scale = rate // 1000000
    cuts, pos = [], 0
    while pos < total:
        kind = rng.integers(0, 4)
        n = int(rng.integers(1, 100)) * 2 if kind == 0 else int(rng.integers(50, 3000)) * 2 * scale if kind == 1 \
            else int(rng.integers(20000, 90000)) * 2 * scale if kind == 2 else int(rng.integers(1, 40)) * 2 * scale * 50
        n = min(n, total - pos, (total // 7) & ~1)
        if n & 1:
            break
        cuts.append(n)
        pos += n
    return cuts


@pytest.mark.parametrize("mode_name,modem,rate,seed", [
    ("2fsk1k", 18, 1000000, 1), ("2fsk1k", 18, 1000000, 2),     # k_decim_pm (one lag tile) + edge scratch, FLL, fused discriminator
    ("gmsk10k", 22, 25000000, 3),                                # k_decim_pm, three lag tiles (25:1)
    ("qpsk250k", 26, 100000000, 4),                              # k_decim_pm at 100:1 + k_dec2_fir + the three-stream pipeline
    ("qpsk250k", 26, 1000000, 5),
    ("2fsk1k", 18, 25000000, 6),                                 # front end, then the 1:50 stage out of a ring (k_decim_pl_gen)
    ("bpsk2k", 0, 2000000, 7), ("4fsk2k", 3, 1000000, 8), ("4fsk100k", 27, 1000000, 9),
])
def test_random_call_sizes(qrl_ctx, mode_name, modem, rate, seed):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(1000 + seed)
    offset = 25000.0 if rate >= 2000000 else 1200.0
    # every second case behind SURVEY 8(d)'s channel (fractional delay, +20 ppm, Es/N0 12 dB): the timing loops' state crosses the random cuts
    # while the symbol phase slides
    iq = sig.make_batch(mode_name, 2, nframes=2, device_rate=rate, rx_offset_hz=offset, seed=40 + seed, impair=sig.SPEC if seed % 2 == 0 else None)
    total = iq.shape[1] & ~1
    cuts = _random_cuts(rng, total, rate)
    used = sum(cuts)
    dem = q.Demod(qrl_ctx, modem, batch=2, max_chunk=max(cuts), device_samp_rate=rate, carrier_offset_hz=offset)
    d = torch.from_numpy(iq).cuda()
    ports = {k: [[], []] for k in ("filtered", "constellation", "bits_a", "bits_b")}
    idx = {"filtered": 0, "constellation": 1, "bits_a": 2, "bits_b": 3}
    pos = 0
    for n in cuts:
        o = dem.process(d[:, pos:pos + n].contiguous())
        c = o["counts"].cpu().numpy()
        for k, j in idx.items():
            host = o[k].cpu().numpy()
            for b in range(2):
                ports[k][b].append(host[b, :c[b, j]].copy())
        pos += n
    dem.close()
    two = not (mode_name.startswith("qpsk") or mode_name.startswith("4fsk"))
    for b in range(2):
        ref = _oracle(mode_name, iq[b, :used], rate, offset)
        for k in ("bits_a", "bits_b"):
            if k == "bits_b" and not two:
                continue
            got = np.concatenate(ports[k][b])
            assert got.size == ref[k].size and np.array_equal(got, ref[k]), (k, b, len(cuts))
        for k in ("filtered", "constellation"):
            got = np.concatenate(ports[k][b]).view(np.float32) + np.float32(0)
            want = ref[k].view(np.float32) + np.float32(0)
            assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, b, len(cuts))
    assert len(cuts) >= 7


@pytest.mark.parametrize("modem,maker,ref", [(41, dict(), "dmr"), (40, dict(alpha=0.5, dev=2400.0), "m17")])
def test_random_call_sizes_4fsk_symbol_demodulators(qrl_ctx, modem, maker, ref):
    import torch
    import qradiolink_amd as q
    rng = np.random.default_rng(modem)
    xs = [sig.make_4fsk(nsym=260, seed=s, **maker)[0] for s in (11, 12)]
    n = min(x.size for x in xs) & ~1
    iq = np.stack([x[:n] for x in xs])
    cuts = _random_cuts(rng, n, 1000000)
    used = sum(cuts)
    dem = q.Demod(qrl_ctx, modem, batch=2, max_chunk=max(cuts))
    d = torch.from_numpy(iq).cuda()
    bits, filt = [[], []], [[], []]
    pos = 0
    for c in cuts:
        o = dem.process(d[:, pos:pos + c].contiguous())
        cn = o["counts"].cpu().numpy()
        for b in range(2):
            bits[b].append(o["bits_a"][b, :cn[b, 2]].cpu().numpy().copy())
            filt[b].append(o["filtered"][b, :cn[b, 0]].cpu().numpy().copy())
        pos += c
    dem.close()
    for b in range(2):
        r = orc.demod_dmr(iq[b, :used]) if ref == "dmr" else orc.demod_m17(iq[b, :used])
        assert np.array_equal(np.concatenate(bits[b]), r["bits_a"])
        assert np.array_equal(np.concatenate(filt[b]).view(np.uint32), r["filtered"].view(np.uint32))
