"""Measurement plumbing of bench.py that can be checked without a GPU: the PMC traffic figure is tied to the kernel sources it was taken
on, the CPU baseline carries SURVEY 8(d)'s thread-per-block variant, and the committed traffic file belongs to the committed sources."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402


def test_committed_pmc_traffic_belongs_to_the_committed_kernel_sources():
    """profiles/pmc_traffic.json is only meaningful for the kernels it was measured on: its source id must be the id of
    qradiolink_amd/csrc + include as they are committed (re-run tools/profile_round.sh + tools/collect_round.py after a kernel change)."""
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    if d["_source_id"] != bench.source_id():      # kernels changed since the pass: bench.py must then report no traffic, with the reason
        t, why = bench.pmc_traffic("c1", "k_decim_pm")
        assert t is None and d["_source_id"] in why
        pytest.skip("profiles/pmc_traffic.json is stale for these kernel sources: re-run tools/profile_round.sh + tools/collect_round.py")
    cal = d["_calibration"]["applied"]
    assert cal["FETCH_SIZE"] == 0.5 and cal["WRITE_SIZE"] == 1.0      # measured in the same pass: tools/pmc_calibrate.py
    for cfg in ("c1", "c2", "c3", "c4", "c5"):
        assert d[cfg]["fetch_bytes"] > 0 and d[cfg]["write_bytes"] > 0 and d[cfg]["kernel"].startswith("qrl::k_")


def test_traffic_is_reported_only_for_matching_sources(monkeypatch):
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    monkeypatch.setattr(bench, "source_id", lambda: d["_source_id"])          # as if run on the sources of the pass
    t, src = bench.pmc_traffic("c1", "k_decim_pm")
    import re
    assert t and t > 35e9 and re.fullmatch(r"profiles/r\d\d_pmc_summary\.txt", src)   # 35.1 GB of algorithmic bytes per launch
    assert bench.pmc_traffic("c1", "k_decim_pm", default_shape=False) == (None, None)
    assert bench.pmc_traffic("c1", "k_some_other_kernel") == (None, None)
    monkeypatch.setattr(bench, "source_id", lambda: "000000000000")
    t, why = bench.pmc_traffic("c1", "k_decim_pm")
    assert t is None and "000000000000" in why


def test_issue_bound_of_the_qpsk_receivers(monkeypatch):
    """round 4: C3 and C5 carry roofline.issue -- VALU wave instructions of one receiver call from the SQ_INSTS_* pass of the same round,
    tied to the same source id, over the RX-alone time measured in the bench."""
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    monkeypatch.setattr(bench, "source_id", lambda: d["_source_id"])
    for cfg, secs in (("c3", 1.93e-3), ("c5", 3.4e-3)):
        r = bench.issue_roofline(cfg, secs)
        top = bench.promote_issue(dict(bound="hbm", achieved=1.0, peak=8000.0, unit="GB/s", frac=0.1, kernel="k_x"), r, "k_qpsk_pipe4", "why")
        assert top["bound"] == "issue" and top["frac"] == r["frac"] and top["kernel"] == "k_qpsk_pipe4" and top["hbm"]["kernel"] == "k_x"   # round 5: the issue side is the headline of these sub-lines
        assert r["bound"] == "issue" and r["peak"] == bench.VALU_ISSUE_PEAK_G and 0.05 < r["frac"] < 1.0
        assert r["valu_wave_instr_per_rx_call"] > 1e8 and "all_classes_frac" not in r and r["all_wave_instr_per_rx_call"] > r["valu_wave_instr_per_rx_call"]   # round 6: no figure that can exceed 1
        assert any("k_fec" in k for k in r["by_kernel"]) and any("k_qpsk_pipe4" in k for k in r["by_kernel"])
        assert not any("k_tx_" in k for k in r["by_kernel"])          # the modulator's kernels are not part of a receiver call
    assert bench.issue_roofline("c5", 3.4e-3, default_shape=False) is None
    monkeypatch.setattr(bench, "source_id", lambda: "000000000000")
    assert bench.issue_roofline("c5", 3.4e-3)["frac"] is None


def test_traffic_close_to_the_algorithmic_bytes_for_the_streaming_front_ends():
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    alg = {"c1": 16384 * 262144 * 8.178, "c2": 384 * 1638400 * 8.033, "c3": 384 * 1638400 * 8.0625}
    for cfg, a in alg.items():
        ratio = (d[cfg]["fetch_bytes"] + d[cfg]["write_bytes"]) / a
        assert 0.98 < ratio < 1.10, (cfg, ratio)
    # C4 (round 5: the record carries the whole chain): the streaming channelizer reads the wideband input once and writes the 64 channel rings once
    # (8 + 8 B per sample); the per-channel kernel re-reads the ring with its 20 % halo and writes int16 + the RRC ring; the chain's sum is what
    # bench.py reports as `traffic` -- two ring hand-offs above the algorithmic 10.3 B per sample
    if "chain_by_kernel" in d["c4"]:
        n = 64 * (1 << 21)
        pfb = [v for k, v in d["c4"]["chain_by_kernel"].items() if k.startswith("k_pfb_stream64")]
        assert pfb and 0.98 < pfb[0] / (n * 16.0) < 1.06
        assert abs(sum(d["c4"]["chain_by_kernel"].values()) - d["c4"]["chain_bytes"]) < 1.0
        assert 3.0 < d["c4"]["chain_bytes"] / (n * bench.C4_BYTES) < 4.5


def test_cpu_baseline_has_the_thread_per_block_variant():
    r = bench.cpu_baseline("c1", 2, budget_s=0.4)
    assert r["kind"] == "port" and r["cores"] == 2 and r["value"] > 0 and r["single_thread"] > 0
    tpb = r["thread_per_block_model"]
    assert tpb["blocks"] >= 10 and tpb["value"] > r["single_thread"] and 0 < tpb["slowest_block_share"] <= 1


def test_source_id_changes_with_the_kernel_sources(tmp_path, monkeypatch):
    a = bench.source_id()
    assert len(a) == 12 and a == bench.source_id()
    # a copy of the tree with one byte appended to a kernel file gives another id
    import shutil
    for d in ("qradiolink_amd/csrc", "include"):
        dst = tmp_path / d
        dst.mkdir(parents=True)
        for fn in os.listdir(os.path.join(ROOT, d)):
            if fn.endswith((".hip", ".cpp", ".hpp", ".h")):
                shutil.copy(os.path.join(ROOT, d, fn), dst / fn)
    with open(tmp_path / "qradiolink_amd/csrc/kernels_ff.hip", "a") as f:
        f.write("\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.source_id() != a


def test_front_end_flop_roofline_on_the_committed_default_line():
    """c1 / c2 / c3 carry the f32 flop roofline of their front-end kernel beside the HBM one (C2 / C3 are compute bound: profiles/r05_sq_front_end_counters.txt);
    on c3 it sits with the HBM figures under roofline.hbm (the issue roofline is that line's headline)"""
    import json
    import os
    import bench
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_bench_default_with_traffic.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    f1, f2, f3 = d["roofline"]["flops"], d["c2"]["roofline"]["flops"], d["c3"]["roofline"]["hbm"]["flops"]
    for f, name in ((f1, "c1"), (f2, "c2"), (f3, "c3")):
        assert f["bound"] == "f32" and f["peak"] == bench.F32_PEAK_TFLOPS and abs(f["algorithmic_flop_per_sample"] - bench.FRONT_END_FLOP[name]) < 0.06
        assert abs(f["frac"] - f["achieved"] / f["peak"]) < 1e-3
    assert 0.35 < f2["frac"] < 0.55 and 0.33 < f3["frac"] < 0.55 and f1["frac"] < 0.2
    assert abs(bench.FRONT_END_FLOP["c2"] - (4 * 1045 / 25 + 6)) < 1e-9
