"""a37b, the DMR DMO correlator slicer (gr_dmr_dmo_sink): the oracle's restatement (oracle/orc_dmr.c) against known answers.
The only things the reference pins are its literal constants: the Golay tables of src/MMDVM/DMRSlotType2.cpp and the sync
patterns of src/DMR/constants.h; both are checked here (the tables against the reference source itself when it is present)."""
import os
import re

import numpy as np
import pytest

import orc
import sig

REF = "/root/reference/src/MMDVM/DMRSlotType2.cpp"


def _ref_table(name):
    text = open(REF).read()
    body = text[text.index(name):]
    body = body[body.index("{") + 1:body.index("}")]
    return np.array([int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]+)U", body)], np.uint32)


def test_golay_tables_match_reference_constants():
    t = orc.golay1987_table()
    # a few literals of DECODING_TABLE_1987 / ENCODING_TABLE_2087 (DMRSlotType2.cpp:46-49, 22-23), always checked
    assert (t[0], t[1], t[15], t[30], t[31]) == (0, 1, 0x24020, 0x48040, 0x01480)
    enc = {1: 0xB08E, 2: 0xE093, 3: 0x501D, 255: 0xD0D6}
    for v, ck in enc.items():   # table entry = {parity byte 1, parity nibble << 4 | ...} of the 20-bit code word
        cw = sig.golay2087_encode(v)
        assert (cw >> 12) == v and ((cw >> 4) & 0xFF) == (ck & 0xFF) and (cw & 0xF) == ((ck >> 12) & 0xF)
    if os.path.exists(REF):     # this container only: the whole 2048-entry table and all 256 code words against the reference source
        assert np.array_equal(t, _ref_table("DECODING_TABLE_1987[]"))
        ref_enc = _ref_table("ENCODING_TABLE_2087[]")
        for v in range(256):
            cw = sig.golay2087_encode(v)
            assert ((cw >> 4) & 0xFF) == (ref_enc[v] & 0xFF) and (cw & 0xF) == ((ref_enc[v] >> 12) & 0xF)


def _same(got, sent):
    """The reference slices 132 symbols starting at m_endPtr - 654 in steps of 5: the LAST one sits at m_endPtr + 1, one sample
    beyond what has been written when the slot is cut (gr_dmr_dmo_sink.cpp:118-122; same arithmetic as MMDVM's DMRDMORX), i.e. it
    is the sample of 1440 samples earlier.  The oracle restates that faithfully, so only the first 131 dibits are the sent ones."""
    return np.array_equal(np.unpackbits(np.frombuffer(got, np.uint8))[:262], np.unpackbits(np.frombuffer(sent, np.uint8))[:262])


def _burst(rng, kinds, cc=5):
    frames = []
    for kind in kinds:
        if kind == "lc":
            frames.append(sig.dmr_frame(rng.integers(0, 2, 196), sig.DMR_MS_DATA_SYNC, cc, 0x01))
        elif kind == "term":
            frames.append(sig.dmr_frame(rng.integers(0, 2, 196), sig.DMR_MS_DATA_SYNC, cc, 0x02))
        elif kind == "csbk":
            frames.append(sig.dmr_frame(rng.integers(0, 2, 196), sig.DMR_MS_DATA_SYNC, cc, 0x03))
        elif kind == "voice_sync":
            frames.append(sig.dmr_frame(rng.integers(0, 2, 216), sig.DMR_MS_VOICE_SYNC))
        else:
            frames.append(sig.dmr_frame(rng.integers(0, 2, 264)))
    return frames


def test_voice_call_is_sliced_into_its_frames():
    """voice LC header, voice superframe (A with sync, B-F without), terminator: every burst comes back as the 33 bytes that
    were sent, typed and numbered like gr_dmr_dmo_sink::processSample does (FN 0 for A, 1..5 for B..F)"""
    rng = np.random.default_rng(1)
    kinds = ["lc", "voice_sync", "v", "v", "v", "v", "v", "voice_sync", "v", "term"]
    frames = _burst(rng, kinds)
    x = sig.dmr_samples(frames) + 0.004 * rng.standard_normal(sig.dmr_samples(frames).size).astype(np.float32)
    got = orc.DmoSink().process(x)
    assert len(got) == len(frames)
    assert all(_same(g[3], f) for g, f in zip(got, frames))
    assert [g[0] for g in got] == [0, 2, 1, 1, 1, 1, 1, 2, 1, 0]          # DMRFrameTypeData / VoiceSync / Voice
    assert [g[1] for g in got] == [0, 0, 1, 2, 3, 4, 5, 0, 1, 0]
    assert all(g[2] == 5 for g in got)


def test_chunked_calls_and_offset_and_polarity():
    """state carries across calls; a DC offset and a different deviation are absorbed by the centre / threshold estimate"""
    rng = np.random.default_rng(2)
    frames = _burst(rng, ["lc", "voice_sync", "v", "v", "csbk"])
    x = 0.7 * sig.dmr_samples(frames, scale=0.5) + 0.11
    whole = orc.DmoSink().process(x)
    snk, parts = orc.DmoSink(), []
    for s in range(0, x.size, 997):
        parts.extend(snk.process(x[s:s + 997]))
    assert parts == whole and all(_same(g[3], f) for g, f in zip(whole, frames)) and len(whole) == 5


def test_sync_with_two_symbol_errors_is_accepted_and_three_rejected():
    rng = np.random.default_rng(3)
    f = bytearray(sig.dmr_frame(rng.integers(0, 2, 196), sig.DMR_MS_DATA_SYNC, 3, 0x03))
    ok, bad = bytearray(f), bytearray(f)
    for k, buf in ((2, ok), (4, bad)):           # flip the sign dibit of k sync symbols (+3 <-> -3: bit 0 of the dibit)
        for j in range(k):
            bit = 108 + 2 * (3 * j + 1)
            buf[bit >> 3] ^= 0x80 >> (bit & 7)
    assert len(orc.DmoSink().process(sig.dmr_samples([bytes(ok)]))) == 1
    assert len(orc.DmoSink().process(sig.dmr_samples([bytes(bad)]))) == 0
