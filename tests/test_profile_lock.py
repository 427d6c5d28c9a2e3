"""tools/profiling/collect_profiles.py refuses a profile set that is not evidence for the bench line:
checked on the committed round-2 set, whose chain / fast-chain kernel stats pre-dated the final kernels."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cp():
    spec = importlib.util.spec_from_file_location("collect_profiles", os.path.join(ROOT, "tools", "profiling", "collect_profiles.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _args(w):
    line = json.loads(open(os.path.join(ROOT, "profiles", f"r02_{w}_large_bench.json")).read())
    trace = os.path.join(ROOT, "profiles", f"r02_{w}_large_kernel_trace.csv")
    return w, os.path.join(ROOT, "profiles", f"r02_{w}_large_kernel_stats.csv"), trace if os.path.exists(trace) else None, line


def test_stale_kernel_stats_are_refused():
    cp = _cp()
    traffic = json.load(open(os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")))
    why = cp.check(*_args("chain"), traffic)
    assert why and "chain_block_kernel<3, false>" in why[0]            # the stats hold chain_block_kernel<3>: an older kernel
    why = cp.check(*_args("fast-chain"), traffic)
    assert any("exceeds the bench line's ms_per_step" in x for x in why)


def test_matching_sets_are_accepted():
    cp = _cp()
    traffic = json.load(open(os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")))
    for w in ("bsw", "bpm", "wfa", "fmi"):
        assert cp.check(*_args(w), traffic) == [], w


def test_a_profile_without_the_priced_kernel_is_refused(tmp_path):
    cp = _cp()
    w, stats, trace, line = _args("wfa")
    other = tmp_path / "stats.csv"
    other.write_text(open(stats).read().replace("wfa_lds_static", "wfa_something_else"))
    assert any("none of the kernels bench.py prices" in x for x in cp.check(w, str(other), trace, line, None))


def test_short_kernel_names():
    cp = _cp()
    assert cp.short("void (anonymous namespace)::chain_block_kernel<3, false>((anonymous namespace)::ChainWork const*, int*)") == "chain_block_kernel<3, false>"
    assert cp.short("__amd_rocclr_copyBuffer") == "__amd_rocclr_copyBuffer"


@pytest.mark.parametrize("rnd", ["r03", "r04"])
def test_a_rounds_committed_set_passes_the_lock(rnd):
    """every workload of the round whose kernel stats and bench line are committed under profiles/ is evidence for that bench
    line: the kernels bench.py prices and those of profiles/<round>_hbm_traffic.json appear in the stats, and average x launches
    per step fits the line's ms_per_step"""
    cp = _cp()
    traffic = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic.json")))
    seen = 0
    for w in ("bsw", "chain", "fast-chain", "bpm", "bitpal", "bitpal-edit", "wfa", "fmi", "fmi-sa", "parse-bsw"):
        stats = os.path.join(ROOT, "profiles", f"{rnd}_{w}_large_kernel_stats.csv")
        bench_line = os.path.join(ROOT, "profiles", f"{rnd}_{w}_large_bench.json")
        if not (os.path.exists(stats) and os.path.exists(bench_line)):
            continue
        trace = os.path.join(ROOT, "profiles", f"{rnd}_{w}_large_kernel_trace.csv")
        line = json.loads(open(bench_line).read().strip().splitlines()[-1])
        assert cp.check(w, stats, trace if os.path.exists(trace) else None, line, traffic) == [], w
        seen += 1
    assert seen >= 6
