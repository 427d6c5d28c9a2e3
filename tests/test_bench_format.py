"""The driver keeps only the tail of bench.py's stdout and parses its LAST line: that line must stay small and complete
(round 2's 21.6 KB line came back as `parsed: null`).  Runs the formatter on the committed payload of a real default run."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "extra")


def _payload():
    """a real result: the r02 default run (headline with the whole suite nested in extra.suite, 21.6 KB)"""
    out = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_default.json")))
    suite = out["extra"].pop("suite")
    out["extra"].pop("suite_note", None)
    out["extra"]["value_hbm_resident"] = out["value"]
    out["extra"]["value_roi_incl_pcie"] = out["extra"]["roi_incl_pcie"]["value"]
    out["cpu_baseline"]["host_threads_visible"] = 256
    for r in suite.values():
        if "value" in r:
            r["extra"]["value_hbm_resident"] = r["value"]
    return out, suite


def test_headline_is_small_complete_json():
    out, suite = _payload()
    assert len(json.dumps(out)) + len(json.dumps(suite)) > 15000          # the payload really is the big one
    line = bench.headline(out, suite)
    assert "\n" not in line and len(line) < 4096
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    assert d["metric"] == out["metric"] and d["value"] == out["value"] and d["ms_per_step"] == out["ms_per_step"]
    assert d["config"]["workload"] == "bsw-large" and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"] and cb["host_threads_visible"] == 256
    assert d["extra"]["value_hbm_resident"] == d["value"] and d["extra"]["value_roi_incl_pcie"] > 0
    # r04: `value` stays the HBM-resident figure (the task's measurement rule), and says so; the comparison with the CPU
    # baseline is like for like -- the drop-in ROI including PCIe against the reference's ROI
    assert d["config"]["value_is"] == "hbm_resident"
    # every suite entry is summarised in the headline
    assert set(d["extra"]["suite"]) == set(suite)
    assert d["extra"]["suite"]["chain-large"][0] == suite["chain-large"]["value"]


def test_x_cpu_baseline_is_like_for_like():
    """extra.x_cpu_baseline = ROI incl. PCIe / CPU baseline; the HBM-resident ratio has its own name; per-core figure and the
    quota are stated (bench.finish_cpu_ratios is what run_one() calls on a fresh result)"""
    out = {"value": 220.0, "extra": {"value_roi_incl_pcie": 170.0}, "cpu_baseline": {"value": 7.75, "cores": 16, "sample": "first 4000000 pairs", "kind": "reference"}}
    bench.finish_cpu_ratios(out, world=1, host_threads=256, quota=16)
    assert out["extra"]["x_cpu_baseline"] == round(170.0 / 7.75, 2) and out["extra"]["x_cpu_baseline_hbm_resident"] == round(220.0 / 7.75, 2)
    assert out["cpu_baseline"]["per_core"] == round(7.75 / 16, 4) and out["cpu_baseline"]["sample"].startswith("16-core cgroup quota of a 256-thread host")
    assert abs(out["extra"]["cpu_cores_equivalent"] - 170.0 / (7.75 / 16)) < 0.2
    out2 = {"value": 220.0, "extra": {}, "cpu_baseline": {"value": 7.75, "cores": 16, "sample": "s", "kind": "reference"}}
    bench.finish_cpu_ratios(out2, world=1, host_threads=16, quota=16)
    assert out2["extra"]["x_cpu_baseline"] is None and out2["extra"]["x_cpu_baseline_hbm_resident"] > 0       # no host ROI measured: no like-for-like figure
    # a one-thread baseline (the line readers) on the same box: the quota is the box's, the threads used are said beside it
    out3 = {"value": 3.0, "extra": {}, "cpu_baseline": {"value": 0.36, "cores": 1, "sample": "s", "kind": "reference"}}
    bench.finish_cpu_ratios(out3, world=1, host_threads=256, quota=16)
    assert out3["cpu_baseline"]["sample"].startswith("16-core cgroup quota of a 256-thread host, 1 thread(s) of it used; ") and out3["cpu_baseline"]["per_core"] == 0.36


def test_emit_prints_suite_lines_then_the_headline_last(tmp_path):
    out, suite = _payload()
    suite["skipped-one"] = {"skipped": "time budget"}
    buf = io.StringIO()
    path = tmp_path / "bench_suite.json"
    with redirect_stdout(buf):
        bench.emit(out, suite, path=str(path))
    lines = buf.getvalue().splitlines()
    assert len(lines) == len(suite) + 1
    for l in lines:
        assert len(l) < 4096
        json.loads(l)
    assert all("suite" in json.loads(l) and isinstance(json.loads(l)["suite"], str) for l in lines[:-1])
    last = json.loads(lines[-1])
    assert last["metric"] == out["metric"] and "roofline" in last and "cpu_baseline" in last
    one = json.loads(lines[0])
    assert one["suite"] == "chain-large" and one["roofline"]["frac"] > 0 and one["cpu_baseline"]["kind"] == "reference"
    full = json.load(open(path))
    assert full["headline"]["extra"]["valu"] and set(full["suite"]) == set(suite)       # nothing is lost, it just moved


def test_headline_sheds_the_suite_rather_than_overflow():
    out, suite = _payload()
    big = {f"entry-{i}": dict(suite["chain-large"]) for i in range(200)}
    line = bench.headline(out, big)
    assert len(line) < 4096
    d = json.loads(line)
    assert "roofline" in d and "cpu_baseline" in d and "suite" not in d["extra"]


@pytest.mark.parametrize("rnd", ["r03", "r04"])
def test_the_committed_default_run_of_a_round_is_a_parseable_record(rnd):
    """profiles/<round>_bench_default_stdout.txt = the stdout of `python bench.py` on the GPU box at the end of the round, as the
    driver reads it: suite lines first, the compact headline LAST"""
    lines = open(os.path.join(ROOT, "profiles", f"{rnd}_bench_default_stdout.txt")).read().splitlines()
    assert all(len(l) < 4096 for l in lines)
    rows = [json.loads(l) for l in lines]
    assert all("suite" in r and isinstance(r["suite"], str) for r in rows[:-1]) and len(rows) >= 12
    last = rows[-1]
    for k in REQUIRED:
        assert k in last, k
    assert last["config"]["workload"] == "bsw-large" and last["n_gpus"] == 1
    assert last["roofline"]["traffic"] and last["cpu_baseline"]["kind"] == "reference" and last["parity"].startswith("bit-exact")
    assert last["extra"]["value_roi_incl_pcie"] and set(last["extra"]["suite"]) >= {"chain-large", "fast-chain-large", "bpm-large", "wfa-large", "fmi-large"}
    for r in rows[:-1]:
        if "value" in r and r["suite"].endswith("-large"):
            assert r["extra"].get("value_roi_incl_pcie"), r["suite"]          # SURVEY.md 8(d)'s ROI for every large workload
    if rnd >= "r04":        # r04: the CPU comparison is like for like, the quota is said in words, `value` says what it is
        assert last["config"]["value_is"] == "hbm_resident" and last["cpu_baseline"]["per_core"] > 0
        assert "cgroup quota" in last["cpu_baseline"]["sample"] or last["cpu_baseline"]["cores"] == last["cpu_baseline"]["host_threads_visible"]
        assert abs(last["extra"]["x_cpu_baseline"] - last["extra"]["value_roi_incl_pcie"] / last["cpu_baseline"]["value"]) < 0.05
