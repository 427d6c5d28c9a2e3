"""CPU (no GPU): BASELINE.json config 1 literally -- the regression harness (benchmarks/*/scripts/regression_small.sh
through benchmarks/run_wrapper.sh) driving the REFERENCE's CPU binaries at one thread: plumbing + output diff.

The scripts take another binary with the same CLI through $GAB_<BENCH>_COMMAND; here that is the compiled reference
under oracle/_ref (test infrastructure, present in the build container only), so the command line the scripts build, the
files they read, the lines they grep and the diff they make are exercised without a GPU.  The same scripts run the
MI355X drivers in tests/test_drivers_gpu.py."""
import os
import shutil
import subprocess

import pytest

from oracle import pyoracle
from tests.make_inputs import make

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REF = {"bsw": ("GAB_BSW_COMMAND", "bsw_ref_avx2"), "chain": ("GAB_CHAIN_COMMAND", "chain_ref"),
       "fast-chain": ("GAB_FASTCHAIN_COMMAND", "fastchain_ref_avx2"), "bpm": ("GAB_BPM_COMMAND", "bpm_ref"),
       "wfa": ("GAB_WFA_COMMAND", "wfa_ref"), "fmi": ("GAB_FMI_COMMAND", "fmi_ref")}

needs_ref = pytest.mark.skipif(pyoracle.ref_path("bsw_ref_avx2") is None, reason="oracle/_ref not built (no /root/reference)")


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    return make(str(tmp_path_factory.mktemp("genarch-inputs")))


def run_script(bench, inputs, cwd, size="small", extra_env=None):
    var, exe = REF[bench]
    env = dict(os.environ, GENARCH_BENCH_INPUTS_ROOT=inputs, **{var: pyoracle.ref_path(exe)}, **(extra_env or {}))
    return subprocess.run(["bash", os.path.join(ROOT, "benchmarks", bench, "scripts", f"regression_{size}.sh")], cwd=cwd, env=env,
                          capture_output=True, text=True, timeout=600)


@needs_ref
@pytest.mark.parametrize("bench", ["bsw", "chain", "fast-chain", "bpm", "wfa", "fmi"])
def test_regression_small_on_reference_cpu_path(inputs, bench, tmp_path):
    # fmi: the reference driver's realloc path dangles on inputs this small at its batch size of 512 (SURVEY.md App. B5:
    # SIGSEGV); $GAB_FMI_BATCH=64 is what the golden fixture was produced with
    r = run_script(bench, inputs, tmp_path, extra_env={"GAB_FMI_BATCH": "64"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout and "FAILED" not in r.stdout, r.stdout
    assert "Kernel execution time" in r.stdout
    assert "_omp_1_gpus_1_" in r.stdout                 # one job: nodes=1, mpi=1, omp=1 (1 thread)


@needs_ref
def test_missing_expected_file_fails_the_job(inputs, tmp_path):
    """the reference's script fails when output-reference.file is absent (bsw/scripts/regression_small.sh:92-96)"""
    broken = str(tmp_path / "inputs")
    shutil.copytree(inputs, broken)
    os.remove(f"{broken}/bsw/small/output-reference.file")
    r = run_script("bsw", broken, tmp_path)
    assert r.returncode != 0 and "FAILED" in r.stdout and "not identical" in r.stdout, r.stdout + r.stderr


@needs_ref
def test_wrong_expected_file_fails_the_job(inputs, tmp_path):
    broken = str(tmp_path / "inputs")
    shutil.copytree(inputs, broken)
    p = f"{broken}/chain/small/out-reference.txt"
    txt = open(p).read().splitlines()
    txt[3] = "0\t-1" if txt[3] != "0\t-1" else "1\t-1"
    open(p, "w").write("\n".join(txt) + "\n")
    r = run_script("chain", broken, tmp_path)
    assert r.returncode != 0 and "FAILED" in r.stdout, r.stdout + r.stderr


def test_invalid_inputs_root_is_an_error(tmp_path):
    env = dict(os.environ, GENARCH_BENCH_INPUTS_ROOT=str(tmp_path / "nowhere"))
    r = subprocess.run(["bash", os.path.join(ROOT, "benchmarks", "bsw", "scripts", "regression_small.sh")], cwd=tmp_path, env=env,
                       capture_output=True, text=True)
    assert r.returncode == 1 and "You have not set a valid input folder" in r.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/benchmarks"), reason="the reference is only in the build container")
def test_reference_citations_resolve():
    """every `file:line` citation of the reference in include/gab.h, the oracle, the library sources, the drivers and the documents
    names a file that exists there and lines it has (tests/check_citations.py)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "check_citations.py")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:]
