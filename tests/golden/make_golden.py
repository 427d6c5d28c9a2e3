#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from THE REFERENCE ITSELF.

Runs only in the build container: needs /root/reference (compiled in place into
oracle/_ref/ by `make -C oracle ref`).  Inputs come from the seeded generators in
tools/gen; expected outputs are what the reference binaries print.  Nothing from
the reference's sources is stored here -- only inputs and the outputs it produced.

    python tests/golden/make_golden.py [bsw ...]
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from tools import gabgen  # noqa: E402

MANIFEST = os.path.join(HERE, "MANIFEST.json")


def _manifest():
    if os.path.exists(MANIFEST):
        return json.load(open(MANIFEST))
    return {}


def _save(m):
    json.dump(m, open(MANIFEST, "w"), indent=1, sort_keys=True)


def make_bsw(m):
    # (name, seed, n, mode): n is a multiple of 32 so no padded pair is printed
    # (bsw/src/main_banded.cpp:254,407-409)
    for name, seed, n, mode in [("bsw_bench", 101, 2048, 0), ("bsw_adv", 102, 1536, 1)]:
        inp = os.path.join(HERE, name + ".in.txt")
        gabgen.write_text("bsw", inp, seed, n, mode)
        outs = {}
        for isa in ("avx512", "avx2", "sse41"):
            exe = pyoracle.ref_path("bsw_ref_" + isa)
            r = subprocess.run([exe, "-pairs", inp, "-t", "1", "-b", "512"], capture_output=True, text=True, check=True)
            outs[isa] = [l for l in r.stderr.splitlines() if "score=" in l][:n]
        assert outs["avx512"] == outs["avx2"] == outs["sse41"], "reference ISA variants disagree"
        with open(os.path.join(HERE, name + ".expected.txt"), "w") as f:
            f.write("\n".join(outs["avx2"]) + "\n")
        # all six fields of the extension result: the driver prints only the score, so the
        # reference's scalarBandedSWAWrapper AND its vector getScores16 are driven by oracle/ref_harness/bsw_full_ref.cpp
        full = {}
        for isa in ("avx512", "avx2"):
            for how in ("scalar", "vector"):
                r = subprocess.run([pyoracle.ref_path("bsw_full_ref_" + isa), inp, how], capture_output=True, text=True, check=True)
                full[isa, how] = r.stdout
        assert len(set(full.values())) == 1, "reference scalar / vector / ISA variants disagree on the full result"
        assert [l.split()[1] for l in full["avx2", "scalar"].splitlines()] == [l.split("=")[1] for l in outs["avx2"]]
        open(os.path.join(HERE, name + ".full.expected.txt"), "w").write(full["avx2", "scalar"])
        m[name] = {"generator": "tools/gen gabgen bsw", "seed": seed, "n": n, "mode": mode,
                   "full_command": "bsw_full_ref_<avx2|avx512> <in> scalar|vector (oracle/ref_harness/bsw_full_ref.cpp on the "
                                   "reference's scalarBandedSWAWrapper and getScores16: identical output) -> '[i] score qle tle gtle gscore max_off'",
                   "reference": "bsw/src/{main_banded,bandedSWA}.cpp built by oracle/Makefile "
                                "(-mavx512bw, -mavx2, -msse4.1: identical output)",
                   "command": "bsw_ref_<isa> -pairs <in> -t 1 -b 512 ; grep score= stderr"}


def make_chain(m):
    # one input serves both benchmarks (fast-chain reads chain's files: fast-chain/scripts/regression_small.sh:4)
    for name, seed, ncalls, mode, nmin, nmax in [("chain_bench", 201, 24, 0, 50, 2000),
                                                 ("chain_dense", 202, 5, 1, 1500, 6000)]:
        inp = os.path.join(HERE, name + ".in.txt")
        gabgen.write_text("chain", inp, seed, ncalls, mode, nmin, nmax)
        outs = {}
        for exe in ("chain_ref", "fastchain_ref_avx2", "fastchain_ref_avx512"):
            out = os.path.join(HERE, name + "." + exe + ".tmp")
            subprocess.run([pyoracle.ref_path(exe), "-i", inp, "-o", out, "-t", "1"], capture_output=True, check=True)
            outs[exe] = open(out).read()
            os.remove(out)
        assert outs["fastchain_ref_avx2"] == outs["fastchain_ref_avx512"], "fast-chain AVX2 != AVX-512"
        open(os.path.join(HERE, name + ".chain.expected.txt"), "w").write(outs["chain_ref"])
        open(os.path.join(HERE, name + ".fastchain.expected.txt"), "w").write(outs["fastchain_ref_avx2"])
        m[name] = {"generator": "tools/gen gabgen chain", "seed": seed, "ncalls": ncalls, "mode": mode,
                   "nmin": nmin, "nmax": nmax,
                   "reference": "chain/src/*.cpp and fast-chain/src/*.cpp built by oracle/Makefile "
                                "(fast-chain: -mavx2 and -mavx512bw builds, identical output)",
                   "command": "<exe> -i <in> -o <out> -t 1"}


def make_bpm(m):
    import re

    def run_ref(inp, alg):
        outs = []
        for t in ("1", "3"):
            out = inp + ".tmp"
            subprocess.run([pyoracle.ref_path("bpm_ref"), "-a", alg, "-i", inp, "-o", out, "-t", t], capture_output=True, check=True)
            lines = sorted(open(out).read().splitlines(), key=lambda l: int(re.match(r"\[(\d+)\]", l).group(1)))
            os.remove(out)
            outs.append(lines)
        assert outs[0] == outs[1], "bpm reference output depends on the thread count"
        return outs[0]

    for name, seed, n, mode, plen in [("bpm_bench", 301, 1500, 0, 151), ("bpm_adv", 302, 2500, 1, 220)]:
        inp = os.path.join(HERE, name + ".in.txt")
        gabgen.write_text("bpm", inp, seed, n, mode, plen)
        open(os.path.join(HERE, name + ".expected.txt"), "w").write("\n".join(run_ref(inp, "bpm-edit")) + "\n")
        # the driver's other two algorithms on the same inputs
        for alg in ("bitpal-edit", "bitpal-scored"):
            open(os.path.join(HERE, f"{name}.{alg.replace('-', '_')}.expected.txt"), "w").write("\n".join(run_ref(inp, alg)) + "\n")
        m[name] = {"generator": "tools/gen gabgen bpm", "seed": seed, "n": n, "mode": mode, "plen": plen,
                   "reference": "bpm/tools/align_benchmark.c + bpm/{benchmark,bitpal,edit,system,utils}/*.c built by oracle/Makefile",
                   "command": "bpm_ref -a bpm-edit|bitpal-edit|bitpal-scored -i <in> -o <out> -t 1|3 ; sort by id (as regression_small.sh:94 does)"}


# adaptive mode: the reduction parameters the adaptive fixtures are produced with
WFA_ADAPTIVE = [(10, 10), (5, 3), (1, 0)]


def make_wfa(m):
    import re

    def run_ref(inp, extra):
        outs = []
        for t in ("1", "3"):
            out = inp + ".tmp"
            subprocess.run([pyoracle.ref_path("wfa_ref"), "-i", inp, "-o", out, "-t", t] + extra, capture_output=True, check=True)
            lines = sorted(open(out).read().splitlines(), key=lambda l: int(re.match(r"id=(\d+)", l).group(1)))
            os.remove(out)
            outs.append(lines)
        assert outs[0] == outs[1], "wfa reference output depends on the thread count"
        return outs[0]

    for name, seed, n, mode, plen in [("wfa_bench", 401, 1500, 0, 151), ("wfa_adv", 402, 2000, 1, 240)]:
        inp = os.path.join(HERE, name + ".in.txt")
        gabgen.write_text("wfa", inp, seed, n, mode, plen)
        open(os.path.join(HERE, name + ".expected.txt"), "w").write("\n".join(run_ref(inp, [])) + "\n")
        m[name] = {"generator": "tools/gen gabgen wfa", "seed": seed, "n": n, "mode": mode, "plen": plen,
                   "reference": "wfa/tools/align_benchmark.c + wfa/{gap_affine,utils}/*.c built by oracle/Makefile",
                   "command": "wfa_ref -i <in> -o <out> -t 1|3 ; sort by id (as regression_small.sh:94 does)"}
    # the same adversarial input through the reference's adaptive reduction
    inp = os.path.join(HERE, "wfa_adv.in.txt")
    for mwl, mdd in WFA_ADAPTIVE:
        lines = run_ref(inp, ["--minimum-wavefront-length", str(mwl), "--maximum-difference-distance", str(mdd)])
        open(os.path.join(HERE, f"wfa_adv.adaptive_{mwl}_{mdd}.expected.txt"), "w").write("\n".join(lines) + "\n")
    m["wfa_adv"]["adaptive_command"] = ("wfa_ref -i <in> -o <out> -t 1|3 --minimum-wavefront-length L --maximum-difference-distance D "
                                        "for (L, D) in " + repr(WFA_ADAPTIVE))


def make_fmi(m):
    # 100 kb reference with planted repeats, 1200 reads of 60..151 bp (2 % substitutions, N's, both strands).
    # batch 64 keeps the reference clear of its realloc / dangling-pointer path.
    name, rseed, qseed, L, n = "fmi_small", 501, 502, 100000, 1200
    ref = gabgen.fmi_ref(rseed, L, 5)
    reads = gabgen.fmi_reads(qseed, ref, n, 60, 151)
    fa = os.path.join(HERE, name + ".ref.fa"); fq = os.path.join(HERE, name + ".reads.fq")
    gabgen.fmi_write_fasta(fa, ref); gabgen.fmi_write_fastq(fq, reads)
    subprocess.run([pyoracle.ref_path("bwa_mem2_index_ref"), "index", fa], capture_output=True, check=True)
    outs = []
    for batch, t in (("64", "1"), ("32", "2")):
        r = subprocess.run([pyoracle.ref_path("fmi_ref"), fa, fq, batch, "19", t], capture_output=True, text=True, check=True)
        lines = r.stdout.splitlines()
        assert not any("realloc" in l for l in lines[:8]), "reference hit its realloc path"
        outs.append("\n".join(lines[6:]) + "\n")          # the harness also drops the 6 header lines
    assert outs[0] == outs[1], "fmi reference output depends on batch size / threads"
    open(os.path.join(HERE, name + ".expected.txt"), "w").write(outs[0])
    # keep only the index header words as a fixture for the index builder (the full file is rebuilt in tests)
    import hashlib
    idx = open(fa + ".bwt.2bit.64", "rb").read()
    m[name] = {"generator": "tools/gen gab_gen_fmi_ref/_reads", "ref_seed": rseed, "read_seed": qseed, "ref_len": L, "n": n,
               "rl_min": 60, "rl_max": 151, "index_sha256": hashlib.sha256(idx).hexdigest(), "index_bytes": len(idx),
               "reference": "fmi/fmi.cpp + bwa-mem2 library, and bwa-mem2 index, built by oracle/Makefile (clang++)",
               "command": "bwa_mem2_index_ref index <fa>; fmi_ref <fa> <fq> 64 19 1 ; drop 6 header lines"}
    # suffix-array look-up: the driver never calls it, so the reference's own
    # FMI_search::get_sa_entries is driven by oracle/ref_harness/fmi_sa_ref.cpp on the SMEMs of this fixture
    import numpy as np
    oidx = pyoracle.fmi_load(fa)
    sm, off = pyoracle.fmi(oidx, reads, 19)
    assert pyoracle.fmi_text(sm, off) == outs[0], "oracle SMEMs differ from the reference's"
    smf = os.path.join(HERE, "_smems.bin"); cof = os.path.join(HERE, "_coords.bin")
    sm.tofile(smf)
    with open(os.path.join(HERE, name + ".sa_expected.txt"), "w") as f:
        for max_occ in (500, 2):
            subprocess.run([pyoracle.ref_path("fmi_sa_ref"), fa, smf, str(max_occ), cof], capture_output=True, check=True)
            co = np.fromfile(cof, np.int64)
            f.write(f"max_occ {max_occ} {len(co)}\n" + "\n".join(str(int(v)) for v in co) + "\n")
    os.remove(smf); os.remove(cof)
    m[name]["sa_command"] = "fmi_sa_ref <fa> <oracle SMEMs of the fixture, 40-byte records> <max_occ> <out> for max_occ 500 and 2"
    for ext in (".0123", ".amb", ".ann", ".pac", ".bwt.2bit.64"):
        os.remove(fa + ext)


MAKERS = {"bsw": make_bsw, "chain": make_chain, "bpm": make_bpm, "wfa": make_wfa, "fmi": make_fmi}

if __name__ == "__main__":
    pyoracle.build(with_ref=True)
    which = sys.argv[1:] or list(MAKERS)
    m = _manifest()
    for w in which:
        MAKERS[w](m)
    _save(m)
    print("golden vectors written:", ", ".join(sorted(m)))
