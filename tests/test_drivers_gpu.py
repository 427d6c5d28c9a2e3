"""GPU: the drop-in C drivers end to end -- reference CLI, reference file formats, reference regression
scripts (benchmarks/*/scripts/regression_small.sh through benchmarks/run_wrapper.sh) against golden outputs."""
import os
import subprocess

import pytest

from tests.make_inputs import make

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "benchmarks"), "-s"])
    return make(str(tmp_path_factory.mktemp("genarch-inputs")))


@pytest.mark.parametrize("bench", ["bsw", "chain", "fast-chain", "bpm", "wfa", "fmi"])
def test_regression_small(inputs, bench, tmp_path):
    env = dict(os.environ, GENARCH_BENCH_INPUTS_ROOT=inputs)
    r = subprocess.run(["bash", os.path.join(ROOT, "benchmarks", bench, "scripts", "regression_small.sh")], cwd=tmp_path,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout and "FAILED" not in r.stdout, r.stdout
    assert "Kernel execution time" in r.stdout


def test_bsw_driver_output_lines(inputs, tmp_path):
    """the lines the reference harness parses are present and well-formed"""
    exe = os.path.join(ROOT, "benchmarks", "bsw", "main_bsw")
    r = subprocess.run([exe, "-pairs", f"{inputs}/bsw/small/bandedSWA_SRR7733443_100k_input.txt", "-t", "1", "-b", "512"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout.startswith("Number of input pairs: 2048\n")
    line = [l for l in r.stdout.splitlines() if l.startswith("Overall SW cycles")][0]
    assert float(line.split(" ")[5]) >= 0          # field 6, as `cut -d ' ' -f 6` takes it
    assert r.stderr.splitlines()[0].startswith("[0] score=")


def test_driver_errors_like_the_reference(tmp_path):
    exe = os.path.join(ROOT, "benchmarks", "bsw", "main_bsw")
    r = subprocess.run([exe, "-t", "1", "-b", "512"], capture_output=True, text=True)
    assert r.returncode != 0 and "pairFileName not specified" in r.stderr
    r = subprocess.run([os.path.join(ROOT, "benchmarks", "fmi", "fmi"), "a", "b"], capture_output=True, text=True)
    assert r.returncode == 1 and "Need five arguments" in r.stdout


def test_bsw_driver_gpu_parse_mode(inputs, tmp_path):
    """GAB_GPU_PARSE=1: the file is parsed on the GPU; stderr (the scores) must be byte-identical
    to the line-by-line mode, and a file the GPU parser declines falls back to it"""
    exe = os.path.join(ROOT, "benchmarks", "bsw", "main_bsw")
    inp = f"{inputs}/bsw/small/bandedSWA_SRR7733443_100k_input.txt"
    a = subprocess.run([exe, "-pairs", inp, "-t", "1", "-b", "512"], capture_output=True, text=True, timeout=300)
    b = subprocess.run([exe, "-pairs", inp, "-t", "1", "-b", "512"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GAB_GPU_PARSE="1", GAB_GPUS="1"))
    assert a.returncode == 0 and b.returncode == 0, b.stderr[-500:]
    assert "input parsed on the GPU" in b.stdout and "input parsed on the GPU" not in a.stdout
    assert a.stderr == b.stderr
    assert [l for l in b.stdout.splitlines() if l.startswith("Overall SW cycles")]
    odd = tmp_path / "odd.txt"
    odd.write_text("123456789\n0123\n012\n" * 32)             # nine-digit h0 line: fgets(temp, 10) splits it
    c = subprocess.run([exe, "-pairs", str(odd), "-t", "1", "-b", "512"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GAB_GPU_PARSE="1", GAB_GPUS="1"))
    assert "GPU parser declined" in c.stderr


@pytest.mark.parametrize("bench", ["chain", "fast-chain"])
def test_chain_driver_gpu_parse_mode(inputs, tmp_path, bench):
    """GAB_GPU_PARSE=1 in the chain drivers: same output file as the fscanf path"""
    exe = os.path.join(ROOT, "benchmarks", bench, "chain")
    inp = f"{inputs}/chain/small/in-1k.txt"
    a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    ra = subprocess.run([exe, "-i", inp, "-o", a, "-t", "1"], capture_output=True, text=True, timeout=300)
    rb = subprocess.run([exe, "-i", inp, "-o", b, "-t", "1"], capture_output=True, text=True, timeout=300,
                        env=dict(os.environ, GAB_GPU_PARSE="1", GAB_GPUS="1"))
    assert ra.returncode == 0 and rb.returncode == 0, rb.stderr[-500:]
    assert "input parsed on the GPU" in rb.stderr and "Time in kernel" in rb.stderr
    assert open(a).read() == open(b).read()


def test_bpm_driver_gpu_parse_mode(inputs, tmp_path):
    """GAB_GPU_PARSE=1 in the bpm driver: same output file as the getline path"""
    exe = os.path.join(ROOT, "benchmarks", "bpm", "bin", "align_benchmark")
    inp = f"{inputs}/bpm/small/BPM_SRR7733443_100k_input.txt"
    a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    ra = subprocess.run([exe, "-a", "bpm-edit", "-i", inp, "-o", a, "-t", "1"], capture_output=True, text=True, timeout=300)
    rb = subprocess.run([exe, "-a", "bpm-edit", "-i", inp, "-o", b, "-t", "1"], capture_output=True, text=True, timeout=300,
                        env=dict(os.environ, GAB_GPU_PARSE="1", GAB_GPUS="1"))
    assert ra.returncode == 0 and rb.returncode == 0, rb.stderr[-500:]
    assert "indexed on the GPU" in rb.stderr
    assert open(a).read() == open(b).read() and len(open(a).read()) > 0


def test_wfa_driver_gpu_parse_mode(inputs, tmp_path):
    """GAB_GPU_PARSE=1 in the wfa driver: same CIGAR file as the getline path"""
    exe = os.path.join(ROOT, "benchmarks", "wfa", "bin", "align_benchmark")
    inp = f"{inputs}/wfa/small/WFA_SRR7733443_100k_input.txt"
    if not os.path.exists(inp):
        inp = [os.path.join(f"{inputs}/wfa/small", f) for f in os.listdir(f"{inputs}/wfa/small") if "input" in f][0]
    a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    ra = subprocess.run([exe, "-i", inp, "-o", a, "-t", "1"], capture_output=True, text=True, timeout=300)
    rb = subprocess.run([exe, "-i", inp, "-o", b, "-t", "1"], capture_output=True, text=True, timeout=300,
                        env=dict(os.environ, GAB_GPU_PARSE="1", GAB_GPUS="1"))
    assert ra.returncode == 0 and rb.returncode == 0, rb.stderr[-500:]
    assert "indexed on the GPU" in rb.stdout
    assert open(a).read() == open(b).read() and len(open(a).read()) > 0


def test_wfa_driver_adaptive_flags(tmp_path):
    """--minimum-wavefront-length / --maximum-difference-distance: the golden CIGARs of the reference's adaptive mode,
    through both the getline path and the GPU parser"""
    from tests.util import GOLDEN
    exe = os.path.join(ROOT, "benchmarks", "wfa", "bin", "align_benchmark")
    want = open(f"{GOLDEN}/wfa_adv.adaptive_5_3.expected.txt").read()
    for env in ({}, {"GAB_GPU_PARSE": "1", "GAB_GPUS": "1"}):
        out = str(tmp_path / "o.txt")
        r = subprocess.run([exe, "-i", f"{GOLDEN}/wfa_adv.in.txt", "-o", out, "--minimum-wavefront-length", "5",
                            "--maximum-difference-distance", "3"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-500:]
        got = sorted(open(out).read().splitlines(), key=lambda l: int(l.split()[0][3:]))
        assert "\n".join(got) + "\n" == want


@pytest.mark.parametrize("alg", ["bitpal-edit", "bitpal-scored"])
def test_bpm_driver_bitpal_algorithms(tmp_path, alg):
    """-a bitpal-edit / bitpal-scored: the golden scores of the reference, through the getline path and the GPU parser"""
    from tests.util import GOLDEN
    exe = os.path.join(ROOT, "benchmarks", "bpm", "bin", "align_benchmark")
    want = open(f"{GOLDEN}/bpm_adv.{alg.replace('-', '_')}.expected.txt").read()
    for env in ({}, {"GAB_GPU_PARSE": "1", "GAB_GPUS": "1"}):
        out = str(tmp_path / "o.txt")
        r = subprocess.run([exe, "-a", alg, "-i", f"{GOLDEN}/bpm_adv.in.txt", "-o", out], capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-500:]
        assert open(out).read() == want
