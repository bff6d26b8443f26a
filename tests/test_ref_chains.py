"""Construction-log pin: the PARAMETERS AND WIRING of the receive chains against the reference's own hier-block constructors.

oracle/_ref/libqrl_rec.so runs the reference's gr_demod_*.cpp constructors, unmodified, against recording stand-ins of the stock GNU
Radio factories (oracle/rec_stub/, oracle/ref_shim_rec.cpp): the log names every stock block the reference creates with the exact
arguments it computed, the firdes call behind every tap vector, and every connect().  The oracle's chains (oracle/orc_chains.c,
orc_analog.c — the restatement the HIP path is bit-exact against) emit the same kind of trace from their primitives
(oracle/orc_trace.c).  For every constructor call site in the reference's mode table (src/gr/gr_demodulator.cpp / gr_modem) the two
must agree: the same connected blocks, each with equal parameters (float parameters compared as the float the GNU Radio signature
narrows them to; filter-design arguments as doubles), and every reference edge respected by the oracle's processing order.

What this does not pin: the stock blocks' arithmetic ([GR-MEM] — GNU Radio is not in the image).  Needs /root/reference (skipped on
the GPU box)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
REC = os.path.join(HERE, "..", "oracle", "_ref", "libqrl_rec.so")
# the reference's logs for every case below, written by tools/make_chain_golden.py through libqrl_rec.so: lets the comparison run where
# /root/reference (hence the library) is absent; test_fixture_is_fresh keeps it equal to the live library where that exists
FIXTURE = os.path.join(HERE, "golden", "ref", "chains.json")
pytestmark = pytest.mark.skipif(not os.path.exists(REC) and not os.path.exists(FIXTURE), reason="neither libqrl_rec.so nor the fixture")

CONST_KIND = {"digital::constellation_bpsk": 0, "digital::constellation_dqpsk": 1, "digital::constellation_rect": 2}

# reference stock block -> oracle primitive; the lambda reorders/drops arguments the oracle's primitive does not take (always checked
# to hold their only supported value)
def _fft(name):
    def f(a):
        assert a[0] == "1", "fft_filter decimation"
        return name, [a[1]]
    return f


def _symsync(name):
    def f(a):
        assert a[6] == "1", "symbol_sync osps"
        assert len(a) == 8 or a[8:] == ["enum:0"], "symbol_sync interpolator (MMSE 8-tap)"
        return name, a[:6] + [a[7]]
    return f


MAP = {
    "filter::rational_resampler_ccf": lambda a: ("resamp_ccf", a),
    "filter::rational_resampler_fff": lambda a: ("resamp_fff", a),
    "filter::fft_filter_ccf": _fft("fir_ccf"),
    "filter::fft_filter_ccc": _fft("fir_ccc"),
    "filter::fft_filter_fff": _fft("fir_fff"),
    "digital::fll_band_edge_cc": lambda a: ("fll_band_edge", a),
    "analog::quadrature_demod_cf": lambda a: ("quad_demod", a),
    "analog::agc2_cc": lambda a: ("agc2_cc", a + ["65536"] if len(a) == 4 else a),
    "analog::agc2_ff": lambda a: ("agc2_ff", a + ["65536"] if len(a) == 4 else a),
    "digital::costas_loop_cc": lambda a: ("costas", a if len(a) == 3 else a + ["false"]),
    "filter::iir_filter_ffd": lambda a: ("iir_ffd", split_args(a[0][1:-1]) + split_args(a[1][1:-1]) + (a[2:] or ["true"])),   # oldstyle defaults to true
    "digital::symbol_sync_ff": _symsync("symbol_sync_ff"),
    "digital::symbol_sync_cc": _symsync("symbol_sync_cc"),
    "digital::diff_phasor_cc": lambda a: ("diff_phasor", a),
    "digital::descrambler_bb": lambda a: ("descramble", a),
    "digital::clock_recovery_mm_cc": lambda a: ("clock_recovery_mm_cc", a),
    "analog::pwr_squelch_cc": lambda a: ("pwr_squelch_cc", a),
    "fec::decoder": lambda a: ("cc_decode_k7", []),
    "fec::encoder": lambda a: ("cc_encode_k7", []),
    "filter::pfb_channelizer_ccf": lambda a: ("pfb_channelizer", a),
    "filter::pfb_synthesizer_ccf": lambda a: ("pfb_synthesizer", a),
    "blocks::rotator_cc": lambda a: ("rotator", a),
    "digital::scrambler_bb": lambda a: ("scramble", a),
    "analog::frequency_modulator_fc": lambda a: ("freq_mod", a),
}

# blocks the oracle's chains restate inline (no primitive of their own); their reference arguments are checked against the constants
# the oracle hard-codes, per chain, in INLINE below
INLINE_KINDS = {"analog::sig_source_f", "blocks::complex_to_mag", "blocks::divide_ff", "blocks::add_const_ff", "analog::rail_ff", "blocks::float_to_complex",
                "blocks::multiply_const_ff", "blocks::multiply_const_cc", "blocks::float_to_uchar", "blocks::delay",
                "blocks::complex_to_float", "blocks::interleave", "blocks::complex_to_real", "blocks::complex_to_mag_squared",
                "blocks::multiply_ff", "blocks::add_ff", "blocks::float_to_short", "analog::phase_modulator_fc",
                "digital::binary_slicer_fb", "blocks::pack_k_bits_bb", "blocks::unpack_k_bits_bb", "digital::map_bb",
                "blocks::packed_to_unpacked_bb", "blocks::repeat", "digital::chunks_to_symbols_bf", "digital::chunks_to_symbols_bc",
                "digital::diff_encoder_bb", "blocks::short_to_float", "blocks::unpacked_to_packed_bb", "blocks::null_sink",
                "blocks::stream_to_streams", "blocks::null_source"}
DOUBLE_PARAMS = {"iir_ffd", "pwr_squelch_cc", "rotator"}          # primitives whose GNU Radio signature takes doubles


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur:
        out.append(cur)
    return out


def parse_call(text):
    m = re.match(r"^([\w:]+)\((.*)\)$", text)
    assert m, text
    return m.group(1), split_args(m.group(2))


class RefGraph:
    def __init__(self, log):
        self.blocks, self.edges, self.calls, self.setter_calls = {}, [], [], None
        for line in log.splitlines():
            if line.startswith("hier "):
                self.name = line[5:]
            elif line.startswith("== "):                      # "<kind>.set_x" logs: what follows is what the setter did
                self.setter_calls = []
            elif "->" in line:
                a, b = line.split(" -> ")
                pa, pb = (int(x) for x in a[1:].split(":")), (int(x) for x in b[1:].split(":"))
                self.edges.append((tuple(pa), tuple(pb)))
            elif re.match(r"^#\d+\.", line):
                (self.calls if self.setter_calls is None else self.setter_calls).append(line)
            else:
                m = re.match(r"^#(\d+) (.*)$", line)
                self.blocks[int(m.group(1))] = parse_call(m.group(2))
        self.connected = {a[0] for a, _ in self.edges} | {b[0] for _, b in self.edges}


def fixture_key(kind, args):
    return kind + "(" + ",".join(str(int(v)) for v in args) + ")"


_seen = {}


def ref_log(kind, *args):
    if not os.path.exists(REC):
        import json
        with open(FIXTURE) as f:
            return json.load(f)[fixture_key(kind, args)]
    _seen[fixture_key(kind, args)] = (kind, args)
    L = C.CDLL(REC)
    L.rr_construct.restype = C.c_char_p
    L.rr_construct.argtypes = [C.c_char_p] + [C.c_int] * 5
    a = list(args) + [0] * (5 - len(args))
    r = L.rr_construct(kind.encode(), *[int(v) for v in a])
    assert r is not None
    return r.decode()


def all_cases():
    """(kind, ctor args) of every parametrised case in this file, by collecting what the tests request"""
    import inspect
    import sys
    mod = sys.modules[__name__]
    out = []
    for name, fn in inspect.getmembers(mod, inspect.isfunction):
        if not name.startswith("test_") or name == "test_fixture_is_fresh":
            continue
        marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
        if not marks:
            combos = [()]
        else:
            combos = [v if isinstance(v, (tuple, list)) else (v,) for v in marks[0].args[1]]
        for c in combos:
            before = set(_seen)
            try:
                fn(*c)
            except AssertionError:
                pass
            out += [_seen[k] for k in set(_seen) - before]
    return out


def oracle_trace(fn, *a, **kw):
    lib = orc._load()
    lib.orc_trace_get.restype = C.c_char_p
    lib.orc_trace_enable(1)
    try:
        fn(*a, **kw)
        return [l for l in lib.orc_trace_get().decode().splitlines() if l]
    finally:
        lib.orc_trace_enable(0)


def norm_value(v, g=None, as_double=False):
    """one argument -> comparable value: numbers as float32 (block parameters) or float64 (design arguments)"""
    v = v.strip()
    if v.startswith("enum:"):
        return float(v[5:])
    if v in ("true", "false"):
        return float(v == "true")
    if v.startswith("#") and g is not None:
        return float(CONST_KIND[g.blocks[int(v[1:])][0]])
    if re.match(r"^[\w]+\(.*\)$", v):                                   # a filter design: compare its arguments as doubles
        name, args = parse_call(v)
        return (name,) + tuple(norm_value(x, g, True) for x in args)
    x = float(v)
    return x if as_double else float(np.float32(x))


def ref_events(g):
    """connected reference blocks -> [(block id, oracle primitive name, normalised args)], plus the inline blocks"""
    ev, inline = [], []
    for bid in sorted(g.connected - {0}):
        kind, args = g.blocks[bid]
        if kind in MAP:
            name, a = MAP[kind](args)
            if name in ("agc2_ff", "agc2_cc"):          # a set_max_gain() call after make() replaces the default of 65536
                for c in g.calls:
                    m = re.match(r"^#%d\.set_max_gain\((.*)\)$" % bid, c)
                    if m:
                        a = a[:4] + [m.group(1)]
            ev.append((bid, name, tuple(norm_value(x, g, name in DOUBLE_PARAMS) for x in a)))
        elif kind in INLINE_KINDS or kind.startswith("custom::"):       # custom:: = the reference's own blocks (pinned in test_ref_blocks.py)
            inline.append((bid, kind, args))
        else:
            raise AssertionError("unmapped reference block %s(%s)" % (kind, ",".join(args)))
    return ev, inline


def fold_soft_quant(g, ev, inline):
    """multiply_const_ff(m) -> add_const_ff(a) -> float_to_uchar is the oracle's soft_quant(m, a)"""
    succ = {}
    for (a, _), (b, _) in g.edges:
        succ.setdefault(a, []).append(b)
    kinds = {bid: (k, a) for bid, k, a in inline}
    used = set()
    for bid, (k, a) in list(kinds.items()):
        if k != "blocks::multiply_const_ff":
            continue
        for n1 in succ.get(bid, []):
            if kinds.get(n1, ("",))[0] == "blocks::add_const_ff":
                for n2 in succ.get(n1, []):
                    if kinds.get(n2, ("",))[0] == "blocks::float_to_uchar":
                        ev.append((n2, "soft_quant", (norm_value(a[0]), norm_value(kinds[n1][1][0]))))
                        used |= {bid, n1, n2}
    return ev, [(b, k, a) for b, k, a in inline if b not in used]


def oracle_events(trace):
    out = []
    for line in trace:
        name, args = parse_call(line)
        out.append((name, tuple(norm_value(x, None, name in DOUBLE_PARAMS) for x in args)))
    return out


def compare(kind, ctor, fn, kw, inline_expect, n=6000, x=None, alias=None, unique=False):
    g = RefGraph(ref_log(kind, *ctor))
    ev, inline = ref_events(g)
    ev, inline = fold_soft_quant(g, ev, inline)
    if alias:       # {(reference primitive, design name): oracle primitive} - a documented restatement choice of the oracle, see the caller
        ev = [(b, alias.get((n_, a[-1][0] if a and isinstance(a[-1], tuple) else None), n_), a) for b, n_, a in ev]
    rng = np.random.default_rng(1)
    if x is None:
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * 0.3
    oev = oracle_events(oracle_trace(fn, x, **kw))
    # 1. the same multiset of (primitive, parameters)
    want = sorted((n_, a) for _, n_, a in ev)
    got = sorted(oev)
    if unique:      # per-channel chains: the reference builds num_channels of them, the oracle one per channelizer output
        want, got = sorted(set(want)), sorted(set(got))
    assert want == got, "\nreference: %s\noracle:    %s" % (want, got)
    # 2. wiring: match each reference block to an oracle event of equal text (in order of first appearance) and require every
    #    reference edge between two mapped blocks to point forward in the oracle's processing order
    pos, taken = {}, set()
    order = topo_order(g)
    for bid in order:
        for b, n_, a in ev:
            if b == bid:
                for i, e in enumerate(oev):
                    if i not in taken and e == (n_, a):
                        pos[bid] = i
                        taken.add(i)
                        break
    reach = mapped_edges(g, set(pos))
    for a, b in reach:
        assert pos[a] < pos[b], "edge #%d -> #%d is not respected by the oracle's order" % (a, b)
    # 3. the inline blocks carry exactly the constants the oracle's chain hard-codes
    got_inline = sorted("%s(%s)" % (k, ",".join(a)) for _, k, a in inline)
    assert got_inline == sorted(inline_expect), got_inline
    return g


def topo_order(g):
    succ, indeg = {}, {}
    for (a, _), (b, _) in g.edges:
        if b == 0 or a == 0:
            continue
        succ.setdefault(a, []).append(b)
        indeg[b] = indeg.get(b, 0) + 1
    nodes = sorted(g.connected - {0})
    ready = [n for n in nodes if indeg.get(n, 0) == 0]
    out = []
    while ready:
        n = ready.pop(0)
        out.append(n)
        for m in succ.get(n, []):
            indeg[m] -= 1
            if indeg[m] == 0:
                ready.append(m)
    assert len(out) == len(nodes), "cycle in the reference graph"
    return out


def mapped_edges(g, mapped):
    """(a, b) for mapped blocks a, b with a path a -> ... -> b through unmapped blocks only"""
    succ = {}
    for (a, _), (b, _) in g.edges:
        if a and b:
            succ.setdefault(a, []).append(b)
    out = set()
    for a in mapped:
        stack, seen = list(succ.get(a, [])), set()
        while stack:
            n = stack.pop()
            if n in seen:
                continue
            seen.add(n)
            if n in mapped:
                out.add((a, n))
            else:
                stack.extend(succ.get(n, []))
    return out


FEC2 = ["blocks::delay(1,1)"]          # the second Viterbi branch's one-item delay (fec_tail)

CASES_2FSK = [(1, 25000, True), (10, 2000, False), (10, 2500, True), (5, 4000, False), (5, 4000, True)]


@pytest.mark.parametrize("sps,fw,fm", CASES_2FSK)
def test_demod_2fsk(sps, fw, fm):
    inline = list(FEC2) + ["blocks::float_to_complex()"]
    if not fm:
        inline += ["blocks::complex_to_mag()", "blocks::complex_to_mag()", "blocks::divide_ff()", "analog::rail_ff(0,2)", "blocks::add_const_ff(-1)"]
    compare("demod_2fsk", (sps, 1000000, 1700, fw, int(fm)), orc.demod_2fsk, dict(sps=sps, filter_width=fw, fm=fm), inline)


@pytest.mark.parametrize("sps,fw", [(1, 20000), (10, 2000), (5, 4000)])
def test_demod_gmsk(sps, fw):
    compare("demod_gmsk", (sps, 1000000, 1700, fw), orc.demod_gmsk, dict(sps=sps, filter_width=fw), FEC2 + ["blocks::float_to_complex()"])


@pytest.mark.parametrize("sps,fw", [(125, 1300), (2, 160000), (25, 6500)])
def test_demod_qpsk(sps, fw):
    compare("demod_qpsk", (sps, 1000000, 1700, fw), orc.demod_qpsk, dict(sps=sps, filter_width=fw),
            ["blocks::multiply_const_cc((-0.707106769,-0.707106769))", "blocks::complex_to_float()", "blocks::interleave(4)"])


@pytest.mark.parametrize("sps,fw", [(10, 1300), (5, 2400)])
def test_demod_bpsk(sps, fw):
    compare("demod_bpsk", (sps, 1000000, 1700, fw), orc.demod_bpsk, dict(sps=sps, filter_width=fw), FEC2 + ["blocks::complex_to_real()"])


def test_demod_m17():
    compare("demod_m17", (125, 1000000, 1700, 9000), orc.demod_m17, dict(),
            ["analog::phase_modulator_fc(1.5707963267948966)", "blocks::complex_to_float()", "blocks::interleave(4)",
             "digital::binary_slicer_fb()", "blocks::pack_k_bits_bb(2)", "digital::map_bb([3,1,2,0])", "blocks::unpack_k_bits_bb(2)"])


@pytest.mark.parametrize("fw", [2500, 5000])
def test_demod_nbfm(fw):
    compare("demod_nbfm", (125, 1000000, 1700, fw), lambda x: orc.demod_analog(x, "nbfm", filter_width=fw), dict(),
            ["blocks::multiply_const_ff(2)"])


def test_demod_nbfm_ctcss_block():
    """gr_demod_nbfm creates _ctcss = ctcss_squelch_ff(8000, 88.5, 0.01, 8000, 160, true) (:59-60) but leaves it unconnected until
    set_ctcss(tone) (:97-123, not a constructor path): the block's constructor parameters equal the ones the oracle's NBFM chain uses
    when the tone is switched on, and the audio filter it swaps in is the band_pass_2 of :112-113 (checked against the source text)."""
    g = RefGraph(ref_log("demod_nbfm", 125, 1000000, 1700, 5000))
    blk = [(b, a) for b, (k, a) in g.blocks.items() if k == "analog::ctcss_squelch_ff"]
    assert len(blk) == 1 and blk[0][0] not in g.connected
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(60000) + 1j * rng.standard_normal(60000)).astype(np.complex64) * 0.3
    tr = [parse_call(l) for l in oracle_trace(lambda v: orc.demod_analog(v, "nbfm", filter_width=5000, ctcss=88.5), x)]
    ct = [a for n_, a in tr if n_ == "ctcss_squelch_ff"]
    assert len(ct) == 1
    assert [norm_value(v) for v in blk[0][1]] == [norm_value(v) for v in ct[0]]
    fir = [a for n_, a in tr if n_ == "fir_fff"]
    assert any("band_pass_2(1,8000,300,3500,200,35," in a[0] for a in fir)
    src = os.path.join("/root/reference/src/gr/gr_demod_nbfm.cpp")
    if os.path.exists(src):
        text = open(src).read().replace(" ", "").replace("\n", "")
        assert "band_pass_2(1,8000,300,3500,200,35,gr::fft::window::WIN_BLACKMAN_HARRIS)" in text
        assert "connect(_audio_resampler,0,_ctcss,0);connect(_ctcss,0,_audio_filter,0);" in text


def test_demod_am():
    compare("demod_am", (125, 1000000, 1700, 5000), lambda x: orc.demod_analog(x, "am", filter_width=5000), dict(),
            ["blocks::complex_to_mag()", "blocks::multiply_const_ff(0.98999999999999999)"], n=20000)


def test_demod_wbfm():
    compare("demod_wbfm", (125, 1000000, 1700, 75000), lambda x: orc.demod_analog(x, "wbfm", filter_width=75000), dict(),
            ["blocks::multiply_const_ff(0.90000000000000002)"], n=20000)


# ---- transmit chains (src/gr/gr_mod_*.cpp; call sites gr_mod_base.cpp) ----
TXBYTES = np.arange(64, dtype=np.uint8)
UNPACK = "blocks::packed_to_unpacked_bb(1,enum:0)"          # MSB first
BB1 = "blocks::multiply_const_cc(1,1)"                      # the bb_gain multiplier at its initial 1


@pytest.mark.parametrize("sps,fw,fm", [(25, 4000, False), (25, 4000, True), (5, 25000, True), (50, 2000, False), (50, 2500, True)])
def test_mod_2fsk(sps, fw, fm):
    inline = [UNPACK, "digital::map_bb([0,1])", "digital::chunks_to_symbols_bf([-1,1])", BB1,
              "blocks::multiply_const_cc(%s,1)" % ("0.899999976" if fm else "0.800000012")]
    if not fm:
        inline.append("blocks::repeat(4,%d)" % sps)
    compare("mod_2fsk", (sps, 1000000, 1700, fw, int(fm)), orc.mod_2fsk, dict(sps=sps, filter_width=fw, fm=fm), inline, x=TXBYTES)


@pytest.mark.parametrize("sps,fw,fm", [(2, 125000, True), (25, 3500, True), (25, 4000, False), (5, 20000, True), (50, 2000, True)])
def test_mod_4fsk(sps, fw, fm):
    inline = [UNPACK, "blocks::pack_k_bits_bb(2)", "digital::map_bb([0,1,3,2])", "digital::chunks_to_symbols_bf([-1.5,-0.5,0.5,1.5])", BB1,
              "blocks::multiply_const_cc(%s,1)" % ("0.899999976" if fm else "0.800000012")]
    if fm:
        inline.append("blocks::multiply_const_ff(0.66666665999999997,1)")
    else:
        inline.append("blocks::repeat(4,%d)" % sps)
    compare("mod_4fsk", (sps, 1000000, 1700, fw, int(fm)), orc.mod_4fsk, dict(sps=sps, filter_width=fw, fm=fm), inline, x=TXBYTES)


@pytest.mark.parametrize("sps,fw", [(10, 20000), (100, 2000), (50, 4000)])
def test_mod_gmsk(sps, fw):
    compare("mod_gmsk", (sps, 1000000, 1700, fw), orc.mod_gmsk, dict(sps=sps, filter_width=fw),
            [UNPACK, "digital::map_bb([0,1])", "digital::chunks_to_symbols_bf([-1,1])", BB1, "blocks::multiply_const_cc(0.899999976,1)"], x=TXBYTES)


@pytest.mark.parametrize("sps,fw", [(100, 6500), (4, 160000), (500, 1300)])
def test_mod_qpsk(sps, fw):
    compare("mod_qpsk", (sps, 1000000, 1700, fw), orc.mod_qpsk, dict(sps=sps, filter_width=fw),
            [UNPACK, "blocks::pack_k_bits_bb(2)", "digital::diff_encoder_bb(4)", "digital::map_bb([0,1,3,2])", BB1,
             "digital::chunks_to_symbols_bc([(-0.707000017,-0.707000017),(-0.707000017,0.707000017),(0.707000017,0.707000017),(0.707000017,-0.707000017)])",
             "blocks::multiply_const_cc(0.59999999999999998,1)"], x=TXBYTES)


def test_mod_m17():
    compare("mod_m17", (125, 1000000, 1700, 9000), orc.mod_m17, dict(),
            [UNPACK, "blocks::pack_k_bits_bb(2)", "digital::map_bb([2,3,1,0])", "digital::chunks_to_symbols_bf([-1.5,-0.5,0.5,1.5])",
             "blocks::multiply_const_ff(0.66666665999999997,1)", "blocks::multiply_const_cc(0.90000000000000002,1)", BB1],
            x=np.arange(48, dtype=np.uint8))


def test_mod_dmr():
    """gr_mod_dmr (src/gr/gr_mod_dmr.cpp:26-90, instance make_gr_mod_dmr() gr_mod_base.cpp:207): the M17 modulator's shape with the DMR pulse
    (RRC(5, 24000, 4800, 0.2, 125)), deviation pi 4800 0.85 / 24000, gr_zero_idle_bursts in the place of the channel filter (the fft_filter_ccf
    the constructor creates is not connected) and the low_pass_2 125 / 3 interpolator at the make() default width 5000"""
    compare("mod_dmr", (), orc.mod_dmr, dict(),
            [UNPACK, "blocks::pack_k_bits_bb(2)", "digital::map_bb([2,3,1,0])", "digital::chunks_to_symbols_bf([-1.5,-0.5,0.5,1.5])",
             "blocks::multiply_const_ff(0.66666665999999997,1)", "custom::gr_zero_idle_bursts()", "blocks::multiply_const_cc(0.90000000000000002,1)", BB1],
            x=np.arange(48, dtype=np.uint8))


@pytest.mark.parametrize("fw", [2500, 5000])
def test_mod_nbfm(fw):
    rng = np.random.default_rng(2)
    compare("mod_nbfm", (20, 1000000, 1700, fw), orc.mod_nbfm, dict(filter_width=fw),
            ["blocks::multiply_const_ff(0.98999999999999999,1)", "blocks::multiply_const_cc(0.80000000000000004,1)", BB1],
            x=(rng.standard_normal(800) * 0.1).astype(np.float32))


def test_mod_nbfm_ctcss_blocks():
    """gr_mod_nbfm creates _tone_source = sig_source_f(8000, GR_COS_WAVE, 88.5, 0.15) and _add = add_ff (:53-54) but leaves them unconnected until
    set_ctcss(tone) (:101-140, not a constructor path): the blocks' constructor parameters are in the construction log, and what set_ctcss does
    -- x0.85 / x0.98, band_pass_2(1, 8000, 300, 3500, 200, 35, BH), tone -> add -> pre-emphasis -- is checked against the source text and
    against the oracle's trace of the tone variant."""
    g = RefGraph(ref_log("mod_nbfm", 20, 1000000, 1700, 5000))
    src_blk = [(b, a) for b, (k, a) in g.blocks.items() if k == "analog::sig_source_f"]
    assert len(src_blk) == 1 and src_blk[0][0] not in g.connected
    assert [norm_value(v) for v in src_blk[0][1][:4]] == [8000.0, norm_value(src_blk[0][1][1]), np.float32(88.5), np.float32(0.15)]
    rng = np.random.default_rng(2)
    x = (rng.standard_normal(800) * 0.1).astype(np.float32)
    tr = [parse_call(l) for l in oracle_trace(lambda v: orc.mod_nbfm(v, filter_width=5000, ctcss=88.5), x)]
    fir = [a for n_, a in tr if n_ == "fir_fff"]
    assert any("band_pass_2(1,8000,300,3500,200,35," in a[0] for a in fir)
    src = "/root/reference/src/gr/gr_mod_nbfm.cpp"
    if os.path.exists(src):
        text = open(src).read().replace(" ", "").replace("\n", "")
        assert "_audio_amplify->set_k(0.85);_audio_filter->set_taps(gr::filter::firdes::band_pass_2(1,target_samp_rate,300,3500,200,35,gr::fft::window::WIN_BLACKMAN_HARRIS));" in text
        assert "_tone_source->set_frequency(value);" in text and "_audio_amplify->set_k(0.98);" in text
        assert "connect(_audio_amplify,0,_add,0);connect(_add,0,_pre_emph_filter,0);connect(_tone_source,0,_add,1);" in text
        assert "sig_source_f::make(target_samp_rate,gr::analog::GR_COS_WAVE,88.5,0.15)" in text


def test_mod_am():
    """gr_mod_am (src/gr/gr_mod_am.cpp:26-74, instance gr_mod_base.cpp:167): agc2_ff with set_max_gain(1), rail, audio band-pass, the
    frequency-0 carrier source added to the audio, 1:125 resampler, gains, the 4545-tap complex band-pass; the feedforward_agc_cc the
    constructor creates is NOT connected and therefore not part of the compared graph"""
    rng = np.random.default_rng(3)
    g = compare("mod_am", (125, 1000000, 1700, 5000), orc.mod_am, dict(),
                ["analog::sig_source_f(8000,enum:102,0,0.5)", "analog::rail_ff(-0.97999999999999998,0.97999999999999998)", "blocks::add_ff()",
                 "blocks::multiply_const_ff(0.94999999999999996,1)", "blocks::float_to_complex()", "blocks::multiply_const_cc(0.5,1)", BB1],
                x=(rng.standard_normal(200) * 0.1).astype(np.float32))
    ffagc = [b for b, (k, _) in g.blocks.items() if k == "analog::feedforward_agc_cc"]
    assert len(ffagc) == 1 and ffagc[0] not in g.connected


@pytest.mark.parametrize("sps,fw", [(250, 2800), (500, 1500)])
def test_mod_bpsk(sps, fw):
    compare("mod_bpsk", (sps, 1000000, 1700, fw), orc.mod_bpsk, dict(sps=sps, filter_width=fw),
            [UNPACK, "digital::chunks_to_symbols_bc([(-1,0),(1,0)])", "blocks::multiply_const_cc(0.59999999999999998,1)", BB1],
            x=np.arange(16, dtype=np.uint8))


@pytest.mark.parametrize("sps,fw,fm", [(1, 20000, True), (10, 2000, True), (2, 125000, True), (5, 3000, True), (5, 4000, False)])
def test_demod_4fsk(sps, fw, fm):
    inline = ["blocks::complex_to_float()", "blocks::interleave(4)"]
    if fm:
        inline.append("analog::phase_modulator_fc(1.5707963267948966)")
    else:
        inline += ["blocks::complex_to_mag()"] * 4 + ["custom::gr_4fsk_discriminator()"]
    compare("demod_4fsk", (sps, 1000000, 1700, fw, int(fm)), orc.demod_4fsk, dict(sps=sps, filter_width=fw, fm=fm), inline, n=8000)


def test_demod_dmr():
    compare("demod_dmr", (5, 1000000), orc.demod_dmr, dict(),
            ["analog::phase_modulator_fc(1.5707963267948966)", "blocks::multiply_const_ff(0.90000000000000002)", "blocks::complex_to_float()",
             "blocks::interleave(4)", "digital::binary_slicer_fb()", "blocks::pack_k_bits_bb(2)", "blocks::unpack_k_bits_bb(2)",
             "digital::map_bb([3,1,2,0])"], n=30000)


def test_demod_dsss():
    compare("demod_dsss", (25, 1000000, 1700, 150), orc.demod_dsss, dict(),
            FEC2 + ["custom::dsss_decoder_cc()", "blocks::complex_to_real()"], n=30000)


@pytest.mark.parametrize("sb", [0, 1])
def test_demod_ssb(sb):
    compare("demod_ssb", (125, 1000000, 1700, 2700, sb), orc.demod_ssb, dict(sb=sb),
            ["blocks::multiply_const_cc(0.90000000000000002)", "blocks::complex_to_real()", "blocks::multiply_const_ff(1.333)",
             "custom::clipper_cc()", "custom::stretcher_cc()"], n=30000)


def test_demod_mmdvm():
    compare("demod_mmdvm", (), orc.demod_mmdvm, dict(),
            ["custom::rssi_tag_block()", "blocks::multiply_const_ff(1)", "blocks::float_to_short(1,32767)"], n=30000)


def test_mod_dsss():
    # the reference shapes the (+-1, 0) chips with a COMPLEX interpolator (real taps); the oracle runs the real interpolator on the real
    # parts (imaginary parts are exactly zero either way) - same taps, same ratio
    compare("mod_dsss", (25, 1000000, 1700, 200), orc.mod_dsss, dict(filter_width=200),
            [UNPACK, "blocks::unpacked_to_packed_bb(1,enum:0)", "digital::chunks_to_symbols_bc([(-1,0),(1,0)])", "custom::dsss_encoder_bb()",
             "blocks::multiply_const_cc(0.65000000000000002,1)", BB1],
            x=np.arange(2, dtype=np.uint8), alias={("resamp_ccf", "root_raised_cosine"): "resamp_fff"})


@pytest.mark.parametrize("fw,sb", [(1000, 0), (2700, 0), (2700, 1)])
def test_mod_ssb(fw, sb):
    rng = np.random.default_rng(3)
    compare("mod_ssb", (125, 1000000, 1700, fw, sb), orc.mod_ssb, dict(filter_width=fw, sb=sb),
            ["blocks::float_to_complex()", "custom::clipper_cc()", "custom::stretcher_cc()", "blocks::multiply_const_cc(0.90000000000000002,1)", BB1],
            x=(rng.standard_normal(4096) * 0.1).astype(np.float32))


def test_mod_mmdvm():
    rng = np.random.default_rng(4)
    compare("mod_mmdvm", (), orc.mod_mmdvm, dict(),
            ["blocks::short_to_float(1,32767)", "blocks::multiply_const_ff(1,1)", "custom::gr_zero_idle_bursts()",
             "blocks::multiply_const_cc(0.80000000000000004,1)", BB1],
            x=(rng.standard_normal(720) * 1000).astype(np.int16))


def test_demod_mmdvm_multi2():
    """the multi-carrier receive graph (configs[3] of BASELINE.json): 10-port channelizer at 250 ksps, then per carrier 24/25 resampler,
    channel filter, RSSI tagger, discriminator, x1, float_to_short into the ZMQ sink"""
    compare("demod_mmdvm_multi2", (7, 25000, 1), lambda x: orc.demod_mmdvm_multi_rssi(x, 10), dict(),
            ["blocks::stream_to_streams(8,10)"] + ["blocks::null_sink(8)"] * 3 + ["custom::gr_mmdvm_sink()"] +
            ["custom::rssi_tag_block()", "blocks::multiply_const_ff(1)", "blocks::float_to_short(1,32767)"] * 7, n=30000, unique=True)


@pytest.mark.parametrize("N", [3, 7])
def test_mod_mmdvm_multi2(N):
    """the multi-carrier transmit graph: per carrier the gr_mod_mmdvm chain at 24 ksps -> 25/24 -> zero_idle_bursts -> 10-port synthesizer"""
    rng = np.random.default_rng(5)
    g = RefGraph(ref_log("mod_mmdvm_multi2", N, 25000, 1))
    lvl = "%.9g" % np.float32(1.0 / N)
    compare("mod_mmdvm_multi2", (N, 25000, 1), orc.mod_mmdvm_multi, dict(),
            ["custom::gr_mmdvm_source()", "blocks::multiply_const_cc(%s)" % lvl, BB1] + ["blocks::null_source(8)"] * (10 - N) +
            ["blocks::short_to_float(1,32767)", "blocks::multiply_const_ff(1,1)", "blocks::multiply_const_cc(0.80000000000000004,1)",
             "custom::gr_zero_idle_bursts()"] * N,
            x=(rng.standard_normal((N, 720)) * 1000).astype(np.int16))


@pytest.mark.parametrize("N", [3, 7])
def test_demod_mmdvm_multi(N):
    """the frequency-translating multi-carrier receive graph (gr_demod_mmdvm_multi): rotator per off-centre carrier, 1:10 decimator, channel
    filter, RSSI tagger, discriminator"""
    compare("demod_mmdvm_multi", (N, 25000, 1), lambda x: orc.demod_mmdvm_xlating(x, N), dict(),
            ["custom::gr_mmdvm_sink()"] + ["custom::rssi_tag_block()", "blocks::multiply_const_ff(1)", "blocks::float_to_short(1,32767)"] * N,
            n=30000)


SETTER_CASES = [
    # kind, constructor arguments + the setter's value, the oracle call with the same setter applied, its input ("iq" / "audio")
    ("demod_nbfm.set_filter_width", (125, 1000000, 1700, 5000, 4000), lambda v: orc.demod_analog(v, "nbfm", filter_width=5000, set_width=4000), "iq"),
    ("demod_nbfm.set_filter_width", (125, 1000000, 1700, 2500, 3000), lambda v: orc.demod_analog(v, "nbfm", filter_width=2500, set_width=3000), "iq"),
    ("demod_am.set_filter_width", (125, 1000000, 1700, 5000, 4000), lambda v: orc.demod_analog(v, "am", filter_width=5000, set_width=4000), "iq"),
    ("demod_wbfm.set_filter_width", (125, 1000000, 1700, 75000, 60000), lambda v: orc.demod_analog(v, "wbfm", filter_width=75000, set_width=60000), "iq"),
    ("demod_usb.set_filter_width", (125, 1000000, 1700, 2700, 2400), lambda v: orc.demod_ssb(v, sb=0, set_width=2400), "iq"),
    ("demod_lsb.set_filter_width", (125, 1000000, 1700, 2700, 2400), lambda v: orc.demod_ssb(v, sb=1, set_width=2400), "iq"),
    ("mod_nbfm.set_filter_width", (20, 1000000, 1700, 5000, 4000), lambda v: orc.mod_nbfm(v, filter_width=5000, set_width=4000), "audio"),
    ("mod_am.set_filter_width", (125, 1000000, 1700, 5000, 4000), lambda v: orc.mod_am(v, filter_width=5000, set_width=4000), "audio"),
    ("mod_usb.set_filter_width", (125, 1000000, 1700, 2700, 2400), lambda v: orc.mod_ssb(v, sb=0, set_width=2400), "audio"),
    ("mod_lsb.set_filter_width", (125, 1000000, 1700, 2700, 2400), lambda v: orc.mod_ssb(v, sb=1, set_width=2400), "audio"),
]


@pytest.mark.parametrize("kind,args,fn,what", SETTER_CASES, ids=[c[0] + "_%d" % c[1][4] for c in SETTER_CASES])
def test_set_filter_width_of_the_analogue_blocks(kind, args, fn, what):
    """gr_demod_base::set_filter_width(width, mode) / gr_mod_base::set_filter_width (src/gr/gr_demod_base.cpp:1155-1185, gr_mod_base.cpp:878-905) forward to
    the blocks' own set_filter_width, which do NOT repeat the constructors' designs (other transition widths, other design functions, a gain of 2 in the SSB
    receiver's audio filter).  The reference's setters run here against the recording stand-ins: every set_taps design they issue must be the design the oracle
    uses for that filter when the same setter is applied, every design of the oracle's chain must come from the constructor or from the setter, and the
    discriminator gain / modulator sensitivity the setter sets must be the oracle's."""
    g = RefGraph(ref_log(kind, *args))
    assert g.setter_calls, "the setter did nothing?"
    rng = np.random.default_rng(5)
    if what == "iq":
        x = ((rng.standard_normal(20000) + 1j * rng.standard_normal(20000)) * 0.3).astype(np.complex64)
    else:
        x = (rng.standard_normal(3000) * 0.1).astype(np.float32)
    tr = [parse_call(l) for l in oracle_trace(fn, x)]
    designs = lambda calls: {norm_value(a, None, True) for _, aa in calls for a in aa if re.match(r"^[a-z_0-9]+\(.*\)$", a.strip())}
    oracle_designs = designs(tr)
    ctor_designs = designs(list(g.blocks.values()))
    set_designs, numbers = set(), []
    for line in g.setter_calls:
        m = re.match(r"^#(\d+)\.(\w+)\((.*)\)$", line)
        assert m, line
        blk, call, arg = int(m.group(1)), m.group(2), m.group(3)
        if call == "set_taps":
            d = norm_value(arg, None, True)
            set_designs.add(d)
            if blk in g.connected:                                   # the SSB blocks set both sideband filters; one of them is in the graph
                assert d in oracle_designs, (line, sorted(oracle_designs))
        else:
            numbers.append((call, float(np.float32(float(arg)))))
    assert oracle_designs <= ctor_designs | set_designs, sorted(oracle_designs - ctor_designs - set_designs)
    # the filters the setter re-designed are no longer the constructor's in the oracle either
    for blk, (k, a) in g.blocks.items():
        replaced = [l for l in g.setter_calls if l.startswith("#%d.set_taps(" % blk)]
        if replaced and blk in g.connected:
            old = designs([(k, a)])
            new = {norm_value(re.match(r"^#\d+\.set_taps\((.*)\)$", replaced[-1]).group(1), None, True)}
            if old != new:
                assert not (old & oracle_designs), (k, a)
    for call, v in numbers:
        name = {"set_gain": "quad_demod", "set_sensitivity": "freq_mod"}[call]
        got = [float(np.float32(float(a[0]))) for n_, a in tr if n_ == name]
        assert got == [v], (call, v, got)


def test_set_gain_of_the_ssb_receiver():
    """gr_demod_base::set_gain (src/gr/gr_demod_base.cpp:1206-1210) -> gr_demod_ssb::set_gain = _if_gain->set_k(value) (gr_demod_ssb.cpp:118-121): the block the
    setter touches is the constructor's multiply_const_cc(0.9) between the resampler and the sideband filter, which is where the oracle applies the new gain"""
    g = RefGraph(ref_log("demod_usb.set_gain", 125, 1000000, 1700, 2700, 500))
    assert len(g.setter_calls) == 1
    m = re.match(r"^#(\d+)\.set_k\((.*)\)$", g.setter_calls[0])
    blk = int(m.group(1))
    assert float(m.group(2)) == 0.5
    k, a = g.blocks[blk]
    assert k == "blocks::multiply_const_cc" and float(np.float32(float(a[0]))) == float(np.float32(0.9))
    src = {s_[0] for s_, d in g.edges if d[0] == blk}
    dst = {d[0] for s_, d in g.edges if s_[0] == blk}
    assert {g.blocks[b][0] for b in src} == {"filter::rational_resampler_ccf"} and {g.blocks[b][0] for b in dst} == {"filter::fft_filter_ccc"}
    rng = np.random.default_rng(6)
    x = ((rng.standard_normal(300000) + 1j * rng.standard_normal(300000)) * 0.3).astype(np.complex64)
    a1, a2 = orc.demod_ssb(x, sb=0), orc.demod_ssb(x, sb=0, gain=0.9)
    assert np.array_equal(a1["filtered"], a2["filtered"]) and np.array_equal(a1["audio"], a2["audio"])
    half = orc.demod_ssb(x, sb=0, gain=0.5)
    assert np.allclose(half["filtered"], a1["filtered"] * np.float32(0.5 / 0.9), rtol=1e-5, atol=1e-7) and not np.array_equal(half["filtered"], a1["filtered"])


@pytest.mark.skipif(not os.path.exists(REC), reason="needs the live library")
def test_fixture_is_fresh():
    import json
    with open(FIXTURE) as f:
        fx = json.load(f)
    if os.environ.get("PYTEST_XDIST_WORKER") and len(_seen) < 50:
        pytest.skip("pytest-xdist spread this file's tests over several workers: the freshness check needs them all in one process")
    for key, (kind, args) in sorted(_seen.items()):
        assert fx.get(key) == ref_log(kind, *args), "stale tests/golden/ref/chains.json (%s): run tools/make_chain_golden.py" % key
    assert len(_seen) >= 50
