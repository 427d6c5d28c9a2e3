"""GPU: the drop-in C drivers end to end -- reference CLI, reference file formats, reference regression
scripts (benchmarks/*/scripts/regression_small.sh through benchmarks/run_wrapper.sh) against golden outputs."""
import os
import subprocess
import sys

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
    a = subprocess.run([exe, "-pairs", inp, "-t", "1", "-b", "512"], capture_output=True, text=True, timeout=300, env=HOST_ENV)
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
    ra = subprocess.run([exe, "-i", inp, "-o", a, "-t", "1"], capture_output=True, text=True, timeout=300, env=HOST_ENV)
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
    ra = subprocess.run([exe, "-a", "bpm-edit", "-i", inp, "-o", a, "-t", "1"], capture_output=True, text=True, timeout=300, env=HOST_ENV)
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
    ra = subprocess.run([exe, "-i", inp, "-o", a, "-t", "1"], capture_output=True, text=True, timeout=300, env=HOST_ENV)
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
    for env in ({"GAB_GPU_PARSE": "0"}, {"GAB_GPU_PARSE": "1", "GAB_GPUS": "1"}):
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
    for env in ({"GAB_GPU_PARSE": "0"}, {"GAB_GPU_PARSE": "1", "GAB_GPUS": "1"}):
        out = str(tmp_path / "o.txt")
        r = subprocess.run([exe, "-a", alg, "-i", f"{GOLDEN}/bpm_adv.in.txt", "-o", out], capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-500:]
        assert open(out).read() == want


# ---- BASELINE.json config 5's code path on one GPU: many chunks, several queue workers per GPU ---------------------
# (benchmarks/common/gab_driver.h: chunks pulled from a shared cursor by GAB_WORKERS_PER_GPU threads per GPU, each with its
# own handle; matches the reference's `omp for schedule(dynamic)` over batches, bsw/src/main_banded.cpp:338-350,
# fmi/fmi.cpp:250-263 with the `rid += batch offset` fix-up of :340-343)
# (GAB_GPU_PARSE=0: the chunk queue is the HOST-pointer path; since r04 the bsw / bpm / wfa drivers parse regular files on the GPU by
# default and run ONE device call per GPU -- test_regression_small and the *_gpu_parse_mode tests cover that)
HOST_ENV = dict(os.environ, GAB_GPU_PARSE="0")          # the line readers + the host-pointer entry points
QUEUE_ENV = {"GAB_WORKERS_PER_GPU": "3", "GAB_QUEUE_REPORT": "1", "GAB_GPU_PARSE": "0"}


def _queue_report(stderr):
    line = [l for l in stderr.splitlines() if l.startswith("gab_queue:")][-1]
    head, counts = line.split(":", 2)[1:]
    nchunks, nworkers = int(head.split()[0]), int(head.split()[3])
    return nchunks, nworkers, [int(c) for c in counts.split()]


@pytest.mark.parametrize("bench,chunk", [("bsw", 100), ("chain", 3000), ("fast-chain", 3000), ("bpm", 117), ("wfa", 64), ("fmi", 50)])
def test_regression_small_many_chunks_three_workers(inputs, bench, chunk, tmp_path):
    """the golden files again, with the input cut into many chunks spread over three workers that share device 0"""
    env = dict(os.environ, GENARCH_BENCH_INPUTS_ROOT=inputs, GAB_CHUNK=str(chunk), clean="0", **QUEUE_ENV)
    r = subprocess.run(["bash", os.path.join(ROOT, "benchmarks", bench, "scripts", "regression_small.sh")], cwd=tmp_path,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout and "FAILED" not in r.stdout, r.stdout


@pytest.mark.parametrize("bench", ["chain", "fast-chain"])
def test_chain_shares_two_logical_gpus(inputs, bench, tmp_path):
    """N GPUs (here: two logical GPUs on one card, GAB_GPU_OVERSUBSCRIBE=1): the calls are dealt longest first to the least-loaded
    GPU (chain/src/host_kernel.cpp:98-105 schedules calls dynamically), every GPU gets ONE gab_chain_run over its
    share, and the output file is the golden one"""
    env = dict(os.environ, GENARCH_BENCH_INPUTS_ROOT=inputs, GAB_GPUS="2", GAB_GPU_OVERSUBSCRIBE="1", clean="0", **QUEUE_ENV)
    r = subprocess.run(["bash", os.path.join(ROOT, "benchmarks", bench, "scripts", "regression_small.sh")], cwd=tmp_path,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout and "FAILED" not in r.stdout, r.stdout
    # the driver's own report: two shares, one call of the library per GPU, balanced within the longest call
    exe = os.path.join(ROOT, "benchmarks", bench, "chain")
    o = str(tmp_path / "o.txt")
    d = subprocess.run([exe, "-i", f"{inputs}/chain/small/in-1k.txt", "-o", o, "-t", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert d.returncode == 0, d.stderr[-500:]
    nchunks, nworkers, counts = _queue_report(d.stderr)
    assert (nchunks, nworkers, counts) == (2, 2, [1, 1])
    sh = [l for l in d.stderr.splitlines() if l.startswith("chain shares")][0].split(":")[1].split()
    calls, anchors, longest = zip(*[tuple(int(v) for v in t.split("/")) for t in sh])
    assert abs(anchors[0] - anchors[1]) <= max(longest) and min(calls) > 0
    one = str(tmp_path / "one.txt")
    r1 = subprocess.run([exe, "-i", f"{inputs}/chain/small/in-1k.txt", "-o", one, "-t", "1"], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and open(o).read() == open(one).read()


@pytest.mark.parametrize("bench", ["bsw", "chain", "fast-chain", "bpm", "wfa"])
@pytest.mark.parametrize("ngpus", [2, 3])
def test_gpu_parse_with_several_gpus(inputs, bench, ngpus, tmp_path):
    """GAB_GPU_PARSE=1 with N GPUs (here N logical GPUs on one card, GAB_GPU_OVERSUBSCRIBE=1): the file is cut at record
    boundaries on the host, every GPU parses ITS piece (gab_*_parse) and runs the kernel on it -- a piece's data never leaves its
    GPU -- and the output is the one-GPU line-by-line driver's, byte for byte (bsw/src/main_banded.cpp:164-206 and
    chain/src/host_data_io.cpp:13-51 are the readers this replaces)"""
    env = dict(os.environ, GAB_GPU_PARSE="1", GAB_GPUS=str(ngpus), GAB_GPU_OVERSUBSCRIBE="1", GAB_QUEUE_REPORT="1")
    a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    if bench == "bsw":
        exe = os.path.join(ROOT, "benchmarks", "bsw", "main_bsw")
        args = [exe, "-pairs", f"{inputs}/bsw/small/bandedSWA_SRR7733443_100k_input.txt", "-t", "1", "-b", "512"]
        ra = subprocess.run(args, capture_output=True, text=True, timeout=300, env=HOST_ENV)
        rb = subprocess.run(args, capture_output=True, text=True, timeout=300, env=env)
        assert ra.returncode == 0 and rb.returncode == 0, rb.stderr[-500:]
        scores = lambda e: [l for l in e.splitlines() if "score=" in l]
        assert scores(ra.stderr) == scores(rb.stderr) and len(scores(ra.stderr)) == 2048
        assert f"on {ngpus} GPU(s) (input parsed on the GPU)" in rb.stdout and rb.stdout.count("] workTicks = ") == ngpus
        report = rb.stderr
    else:
        if bench in ("chain", "fast-chain"):
            exe = os.path.join(ROOT, "benchmarks", bench, "chain")
            mk = lambda o: [exe, "-i", f"{inputs}/chain/small/in-1k.txt", "-o", o, "-t", "1"]
        else:
            exe = os.path.join(ROOT, "benchmarks", bench, "bin", "align_benchmark")
            name = "BPM" if bench == "bpm" else "WFA"
            inp = f"{inputs}/{bench}/small/{name}_SRR7733443_100k_input.txt"
            mk = lambda o: [exe] + (["-a", "bpm-edit"] if bench == "bpm" else []) + ["-i", inp, "-o", o, "-t", "1"]
        ra = subprocess.run(mk(a), capture_output=True, text=True, timeout=300, env=HOST_ENV)
        rb = subprocess.run(mk(b), capture_output=True, text=True, timeout=300, env=env)
        assert ra.returncode == 0 and rb.returncode == 0, rb.stderr[-500:]
        assert "on the GPU" in rb.stderr + rb.stdout
        key = (lambda l: int(l.split()[0][3:])) if bench == "wfa" else None
        ta, tb = open(a).read(), open(b).read()
        if bench == "wfa":
            ta, tb = sorted(ta.splitlines(), key=key), sorted(tb.splitlines(), key=key)
        elif bench == "bpm":
            ta, tb = sorted(ta.splitlines()), sorted(tb.splitlines())
        assert ta == tb and len(ta) > 0
        report = rb.stderr
    line = [l for l in report.splitlines() if l.startswith("gab GPU parse:")][0]
    per = [int(v) for v in line.split(":")[2].split()]
    assert len(per) == ngpus and all(v > 0 for v in per) and f"{ngpus} piece(s)" in line


def test_queue_spreads_chunks_over_workers(inputs, tmp_path):
    """every chunk runs exactly once, on some worker; more than one worker takes part"""
    exe = os.path.join(ROOT, "benchmarks", "bsw", "main_bsw")
    inp = f"{inputs}/bsw/small/bandedSWA_SRR7733443_100k_input.txt"
    one = subprocess.run([exe, "-pairs", inp, "-t", "1", "-b", "512"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, GAB_WORKERS_PER_GPU="1", GAB_GPU_PARSE="0"))
    many = subprocess.run([exe, "-pairs", inp, "-t", "1", "-b", "512"], capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, GAB_CHUNK="64", **QUEUE_ENV))
    assert one.returncode == 0 and many.returncode == 0, many.stderr[-500:]
    nchunks, nworkers, counts = _queue_report(many.stderr)
    assert nchunks == 2048 // 64 and nworkers == 3 and sum(counts) == nchunks
    assert sum(c > 0 for c in counts) >= 2
    scores = lambda e: [l for l in e.splitlines() if "score=" in l]
    assert scores(one.stderr) == scores(many.stderr) and len(scores(one.stderr)) == 2048
    assert many.stdout.count("] workTicks = ") == 3


def test_fmi_chunks_fix_up_read_ids(inputs, tmp_path):
    """fmi: chunk-local read ids are shifted by the chunk's first read (fmi/fmi.cpp:340-343) and the chunks are printed in
    order: stdout after the six header lines is identical whatever the chunking"""
    exe = os.path.join(ROOT, "benchmarks", "fmi", "fmi")
    args = [exe, f"{inputs}/fmi/broad", f"{inputs}/fmi/small/SRR7733443_1m_1.fastq", "512", "19", "1"]
    a = subprocess.run(args, capture_output=True, text=True, timeout=300, env=dict(os.environ, GAB_WORKERS_PER_GPU="1"))
    b = subprocess.run(args, capture_output=True, text=True, timeout=300, env=dict(os.environ, GAB_CHUNK="37", **QUEUE_ENV))
    assert a.returncode == 0 and b.returncode == 0, b.stderr[-500:]
    body = lambda o: o.splitlines()[6:]
    assert body(a.stdout) == body(b.stdout) and len(body(a.stdout)) > 1200
    assert body(b.stdout) == open(f"{inputs}/fmi/small/out-reference.txt").read().splitlines()
    nchunks, nworkers, counts = _queue_report(b.stderr)
    assert nchunks == (1200 + 36) // 37 and sum(counts) == nchunks


def test_fmi_driver_worker_arrays_fill_up(inputs, tmp_path):
    """the workers collect their SMEMs in page-locked arrays of their own (the reference's per-thread matchArray,
    fmi/fmi.cpp:243-255) and take a separate block for a chunk that no longer fits (:277-286): same output either way,
    also from pageable arrays"""
    exe = os.path.join(ROOT, "benchmarks", "fmi", "fmi")
    args = [exe, f"{inputs}/fmi/broad", f"{inputs}/fmi/small/SRR7733443_1m_1.fastq", "512", "19", "1"]
    want = open(f"{inputs}/fmi/small/out-reference.txt").read().splitlines()
    for extra in ({"GAB_FMI_ARENA": "300"}, {"GAB_FMI_ARENA": "1"}, {"GAB_NO_PIN": "1"}):
        r = subprocess.run(args, capture_output=True, text=True, timeout=300, env=dict(os.environ, GAB_CHUNK="100", **QUEUE_ENV, **extra))
        assert r.returncode == 0, r.stderr[-500:]
        assert r.stdout.splitlines()[6:] == want, extra


def test_wfa_driver_packed_and_unpacked_output(inputs, tmp_path):
    """the wfa driver gets the printed CIGAR text from the device (gab_wfa_run_packed); GAB_WFA_UNPACKED=1 takes the
    operations (gab_wfa_run) and encodes on the host: same file, also when a chunk's text outgrows its room (very
    divergent pairs: the chunk gets a block of its own)"""
    import numpy as np
    exe = os.path.join(ROOT, "benchmarks", "wfa", "bin", "align_benchmark")
    rng = np.random.default_rng(5)
    src = tmp_path / "divergent.txt"
    with open(src, "wb") as f:
        for i in range(3000):
            n = int(rng.integers(20, 150))
            p = rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes()
            t = rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes() if i % 2 else p       # random text: a run per operation
            f.write(b">" + p + b"\n<" + t + b"\n")
    outs = []
    # ({}: the default -- the file indexed on the GPU, gab_wfa_run_packed_device, with the same too-little-room retry)
    for env in ({}, {"GAB_GPU_PARSE": "0"}, {"GAB_GPU_PARSE": "0", "GAB_WFA_UNPACKED": "1"}, {"GAB_CHUNK": "64", **QUEUE_ENV}):
        o = tmp_path / f"o{len(outs)}.txt"
        r = subprocess.run([exe, "-i", str(src), "-o", str(o)], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-500:]
        outs.append(open(o).read())
    key = lambda t: sorted(t.splitlines(), key=lambda l: int(l.split()[0][3:]))
    assert key(outs[0]) == key(outs[1]) and outs[1] == outs[2] == outs[3] and outs[0].count("\n") == 3000


def test_unpinned_and_piped_inputs(inputs, tmp_path):
    """GAB_NO_PIN=1 (pageable slabs) gives the same scores; a pipe instead of a file is refused by the seek the reference
    needs too, and GAB_GPU_PARSE on a pipe never reads an unknown size"""
    exe = os.path.join(ROOT, "benchmarks", "bsw", "main_bsw")
    inp = f"{inputs}/bsw/small/bandedSWA_SRR7733443_100k_input.txt"
    a = subprocess.run([exe, "-pairs", inp, "-t", "1", "-b", "512"], capture_output=True, text=True, timeout=300)
    b = subprocess.run([exe, "-pairs", inp, "-t", "1", "-b", "512"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GAB_NO_PIN="1", GAB_GPU_PARSE="0"))
    assert a.returncode == 0 and b.returncode == 0 and a.stderr == b.stderr
    for bench, args in [("bpm/bin/align_benchmark", ["-a", "bpm-edit", "-o", str(tmp_path / "o.txt")]),
                        ("wfa/bin/align_benchmark", ["-o", str(tmp_path / "o.txt")])]:
        name = "BPM" if bench.startswith("bpm") else "WFA"
        src = f"{inputs}/{bench.split('/')[0]}/small/{name}_SRR7733443_100k_input.txt"
        want = str(tmp_path / "want.txt")
        r0 = subprocess.run([os.path.join(ROOT, "benchmarks", bench)] + args[:-1] + [want, "-i", src], capture_output=True, text=True, timeout=300)
        assert r0.returncode == 0, r0.stderr[-300:]
        # process substitution: the driver reads a pipe; with GAB_GPU_PARSE=1 it must take the getline path
        cmd = f"{os.path.join(ROOT, 'benchmarks', bench)} {' '.join(args)} -i <(cat {src})"
        r = subprocess.run(["bash", "-c", cmd], capture_output=True, text=True, timeout=300, env=dict(os.environ, GAB_GPU_PARSE="1", GAB_GPUS="1"))
        assert r.returncode == 0, r.stderr[-300:]
        assert "indexed on the GPU" not in r.stderr + r.stdout
        assert open(str(tmp_path / "o.txt")).read() == open(want).read()


def test_perf_analysis_fifo_protocol(inputs, tmp_path):
    """-DPERF_ANALYSIS=1: the driver opens perf_ctl.fifo in its working directory and writes "enable" before and "disable"
    after the ROI (bsw/src/main_banded.cpp:313-321,356-359); the reader here plays `perf stat --control fifo:`"""
    import threading
    build = tmp_path / "bin"
    build.mkdir()
    src = os.path.join(ROOT, "benchmarks", "bsw", "src", "main_banded.c")
    lib = os.path.join(ROOT, "genarchbench_amd")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-I", os.path.join(ROOT, "include"), "-DPERF_ANALYSIS=1", "-DPWR=1", src, "-o",
                           str(build / "main_bsw_perf"), "-L", lib, "-lgab_hip", f"-Wl,-rpath,{lib}", "-lpthread", "-lm", "-ldl"])
    fifo = tmp_path / "perf_ctl.fifo"
    os.mkfifo(fifo)
    got = []

    def reader():
        with open(fifo, "rb") as f:
            got.append(f.read())
    t = threading.Thread(target=reader)
    t.start()
    r = subprocess.run([str(build / "main_bsw_perf"), "-pairs", f"{inputs}/bsw/small/bandedSWA_SRR7733443_100k_input.txt", "-t", "1",
                        "-b", "512"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    t.join(timeout=30)
    assert r.returncode == 0, r.stderr[-500:]
    assert got and got[0] == b"enabledisable"
    assert "ERROR opening the Perf pipe" not in r.stderr
    # -DPWR=1: the energy line of the reference's harness (grep "Energy consumption:"), when ROCm SMI is readable
    e = [l for l in r.stdout.splitlines() if l.startswith("Energy consumption:")]
    assert len(e) <= 1 and all(float(l.split()[2]) >= 0 for l in e)
    assert open(f"{inputs}/bsw/small/output-reference.file").read().splitlines() == [l for l in r.stderr.splitlines() if "score=" in l]


def test_default_read_phase(inputs, tmp_path):
    """r04: with GAB_GPU_PARSE unset all the drivers parse a regular file on the GPU and keep the input there (the region of
    interest is kernels + results back; chain's DP kernel writes its results through to the host arrays);
    GAB_GPU_PARSE=0 turns the GPU parsers off everywhere"""
    base = {k: v for k, v in os.environ.items() if k != "GAB_GPU_PARSE"}
    runs = [("bsw", [os.path.join(ROOT, "benchmarks", "bsw", "main_bsw"), "-pairs", f"{inputs}/bsw/small/bandedSWA_SRR7733443_100k_input.txt", "-t", "1", "-b", "512"], True),
            ("bpm", [os.path.join(ROOT, "benchmarks", "bpm", "bin", "align_benchmark"), "-a", "bpm-edit", "-i", f"{inputs}/bpm/small/BPM_SRR7733443_100k_input.txt", "-o", str(tmp_path / "b.txt")], True),
            ("wfa", [os.path.join(ROOT, "benchmarks", "wfa", "bin", "align_benchmark"), "-i", f"{inputs}/wfa/small/WFA_SRR7733443_100k_input.txt", "-o", str(tmp_path / "w.txt")], True),
            ("chain", [os.path.join(ROOT, "benchmarks", "chain", "chain"), "-i", f"{inputs}/chain/small/in-1k.txt", "-o", str(tmp_path / "c.txt"), "-t", "1"], True)]
    for name, args, on_gpu in runs:
        r = subprocess.run(args, capture_output=True, text=True, timeout=300, env=base)
        assert r.returncode == 0, (name, r.stderr[-300:])
        assert ("on the GPU" in r.stdout + r.stderr.split("score=")[0]) == on_gpu, name
        r0 = subprocess.run(args, capture_output=True, text=True, timeout=300, env=dict(base, GAB_GPU_PARSE="0"))
        assert r0.returncode == 0 and "on the GPU" not in r0.stdout + r0.stderr.split("score=")[0], name


@pytest.mark.gpu
def test_pageable_arrays_staged_by_the_library(tmp_path):
    """GAB_STAGE_PAGEABLE=1 (gab_internal.h: gab_memcpy / gab_memcpy_async): copies of more than 1 MiB from or to pageable memory go
    through the library's own page-locked buffers in 8 MiB pieces; results identical to the runtime's own way, for arrays that are not
    a multiple of the piece (bsw 300 k pairs, wfa 80 k pairs with 12 MB of operation room, chain 700 k anchors, both directions)"""
    import numpy as np
    code = (
        "import sys, numpy as np\n"
        "from tools import gabgen\n"
        "from genarchbench_amd.bsw import BandedPairWiseSW\n"
        "from genarchbench_amd.wfa import AffineWavefronts\n"
        "from genarchbench_amd.chain import ChainEngine\n"
        "out = {}\n"
        "sw = BandedPairWiseSW(); out['bsw'] = sw.getScores16(gabgen.bsw(61, 300000, 0)); sw.close()\n"
        "w = AffineWavefronts(); ops, off, ln, sc = w.align(gabgen.pairs(62, 80000, 1, 151)); w.close()\n"
        "out['wfa_ops'] = ops; out['wfa_len'] = ln; out['wfa_score'] = sc\n"
        "c = ChainEngine(); b = gabgen.chain(63, 700, 0, 500, 1500)\n"
        "for m in (0, 1):\n"
        "    s, p = c.host_chain_kernel(b, m); out[f'chain_s{m}'] = s; out[f'chain_p{m}'] = p\n"
        "c.close()\n"
        "np.savez(sys.argv[1], **out)\n")
    got = {}
    for name, stage in (("staged", "1"), ("plain", "0")):
        path = tmp_path / f"{name}.npz"
        env = dict(os.environ, GAB_STAGE_PAGEABLE=stage, PYTHONPATH=ROOT)
        subprocess.run([sys.executable, "-c", code, str(path)], cwd=ROOT, env=env, check=True, timeout=300)
        got[name] = np.load(path)
    assert sorted(got["staged"].files) == sorted(got["plain"].files) and len(got["plain"].files) == 8
    for k in got["plain"].files:
        assert got["plain"][k].size > 0
        np.testing.assert_array_equal(got["staged"][k], got["plain"][k], err_msg=k)
    assert got["plain"]["bsw"].nbytes > (1 << 20) and got["plain"]["wfa_ops"].nbytes > (8 << 20)
