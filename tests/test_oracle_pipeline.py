"""The emulated thread-per-block scheduler of bench.py's cpu_baseline (oracle/orc_pipeline.c) decodes what the plain chain decodes."""
import numpy as np

import orc
import sig


def test_thread_per_block_pipeline_equals_the_chain():
    iq = sig.make_batch("2fsk1k", 5, nframes=2, device_rate=1000000, rx_offset_hz=1200.0, seed=21, impair=sig.SPEC)
    secs, chk = orc.batch_rx(orc.MODE_2FSK_1K if hasattr(orc, "MODE_2FSK_1K") else 0, iq, 1000000, 1200.0, 2)
    psecs, pchk, busy = orc.pipeline_rx_2fsk1k(iq, 1200.0)
    assert pchk == chk and chk != 0
    assert len(busy) == 11 and all(b > 0 for b in busy)
    # the pipeline cannot be faster than its slowest stage allows, nor slower than the stages run one after the other (+ slack for a loaded host)
    assert max(busy) <= psecs * 1.05 and psecs <= sum(busy) * 1.5 + 0.5
