"""CPU: the fmi oracle (oracle/fmi.c) and the index builder (tools/mkindex) against the golden output of the
compiled reference (fmi driver + bwa-mem2 index)."""
import hashlib
import json
import subprocess

import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen, mkindex
from tests.util import GOLDEN, read_fasta_codes, read_fastq_reads

MAN = json.load(open(f"{GOLDEN}/MANIFEST.json"))["fmi_small"]


@pytest.fixture(scope="module")
def index_prefix(tmp_path_factory):
    d = tmp_path_factory.mktemp("fmi")
    ref = read_fasta_codes(f"{GOLDEN}/fmi_small.ref.fa")
    idx = mkindex.FmIndex(ref)
    prefix = str(d / "ref.fa")
    idx.write(prefix)
    return prefix


def test_mkindex_reproduces_reference_index_file(index_prefix):
    """byte-identical to what the reference's `bwa-mem2 index` wrote for the same FASTA (hash in the manifest)"""
    data = open(index_prefix + ".bwt.2bit.64", "rb").read()
    assert len(data) == MAN["index_bytes"]
    assert hashlib.sha256(data).hexdigest() == MAN["index_sha256"]


def test_generators_match_fixture_files():
    ref = gabgen.fmi_ref(MAN["ref_seed"], MAN["ref_len"], 5)
    np.testing.assert_array_equal(ref, read_fasta_codes(f"{GOLDEN}/fmi_small.ref.fa"))
    reads = gabgen.fmi_reads(MAN["read_seed"], ref, MAN["n"], MAN["rl_min"], MAN["rl_max"])
    fq = read_fastq_reads(f"{GOLDEN}/fmi_small.reads.fq")
    np.testing.assert_array_equal(reads.len, fq.len)
    np.testing.assert_array_equal(reads.enc, fq.enc)


def test_oracle_matches_golden(index_prefix):
    idx = pyoracle.fmi_load(index_prefix)
    reads = read_fastq_reads(f"{GOLDEN}/fmi_small.reads.fq")
    sm, off = pyoracle.fmi(idx, reads, 19)
    assert pyoracle.fmi_text(sm, off) == open(f"{GOLDEN}/fmi_small.expected.txt").read()


def test_smem_properties(index_prefix):
    """size-independent: every reported interval is an exact match of the read in the indexed text with
    exactly s occurrences (counted by brute force on a small reference)"""
    idx = pyoracle.fmi_load(index_prefix)
    ref = read_fasta_codes(f"{GOLDEN}/fmi_small.ref.fa")
    both = np.concatenate([ref, 3 - ref[::-1]]).astype(np.uint8).tobytes()
    reads = read_fastq_reads(f"{GOLDEN}/fmi_small.reads.fq")
    sm, off = pyoracle.fmi(idx, reads, 19)
    for j in range(0, len(sm), 97):
        r = sm[j]
        sub = reads.enc[r["rid"], r["m"]:r["n"] + 1].tobytes()
        cnt, pos = 0, both.find(sub)
        while pos >= 0:
            # occurrences spanning the forward / reverse-complement junction are not in the index text
            cnt += 1
            pos = both.find(sub, pos + 1)
        assert cnt >= r["s"] >= 1 and cnt - r["s"] <= 1


@pytest.mark.skipif(pyoracle.ref_path("fmi_ref") is None, reason="oracle/_ref not built (no /root/reference)")
def test_oracle_matches_live_reference(tmp_path):
    ref = gabgen.fmi_ref(71, 60000, 10)
    reads = gabgen.fmi_reads(72, ref, 800, 30, 200)
    fa = str(tmp_path / "r.fa"); fq = str(tmp_path / "q.fq")
    gabgen.fmi_write_fasta(fa, ref); gabgen.fmi_write_fastq(fq, reads)
    subprocess.run([pyoracle.ref_path("bwa_mem2_index_ref"), "index", fa], capture_output=True, check=True)
    r = subprocess.run([pyoracle.ref_path("fmi_ref"), fa, fq, "32", "19", "1"], capture_output=True, text=True, check=True)
    lines = r.stdout.splitlines()
    assert not any("realloc" in l for l in lines[:8])
    want = "\n".join(lines[6:]) + "\n"
    # index from our own builder, not the reference's file
    mkindex.FmIndex(ref).write(str(tmp_path / "mine"))
    idx = pyoracle.fmi_load(str(tmp_path / "mine"))
    sm, off = pyoracle.fmi(idx, reads, 19)
    assert pyoracle.fmi_text(sm, off) == want


# ---- suffix-array look-up -------------------------------------------------------------------
def _sa_golden():
    out, cur = {}, None
    for line in open(f"{GOLDEN}/fmi_small.sa_expected.txt"):
        if line.startswith("max_occ"):
            _, mo, n = line.split(); cur = out.setdefault(int(mo), [])
        else:
            cur.append(int(line))
    return {k: np.array(v, np.int64) for k, v in out.items()}


def test_sa_lookup_matches_reference(index_prefix):
    """oracle_fmi_sa_lookup == the reference's FMI_search::get_sa_entries (driven by oracle/ref_harness) on the
    SMEMs of the fixture, for max_occ 500 (every row) and 2 (strided rows)"""
    idx = pyoracle.fmi_load(index_prefix)
    reads = read_fastq_reads(f"{GOLDEN}/fmi_small.reads.fq")
    sm, _ = pyoracle.fmi(idx, reads, 19)
    for max_occ, want in _sa_golden().items():
        coords, off, _steps = pyoracle.fmi_sa_lookup(idx, sm, max_occ)
        np.testing.assert_array_equal(coords, want)
        assert off[-1] == len(want)


def test_sa_coordinates_are_occurrences(index_prefix):
    """size-independent property: every coordinate is a position where the seed occurs in the indexed text
    (forward strand followed by its reverse complement)"""
    idx = pyoracle.fmi_load(index_prefix)
    ref = read_fasta_codes(f"{GOLDEN}/fmi_small.ref.fa")
    both = np.concatenate([ref, 3 - ref[::-1]]).astype(np.uint8)
    reads = read_fastq_reads(f"{GOLDEN}/fmi_small.reads.fq")
    sm, _ = pyoracle.fmi(idx, reads, 19)
    coords, off, _ = pyoracle.fmi_sa_lookup(idx, sm, 500)
    for j in range(0, len(sm), 53):
        r = sm[j]
        sub = reads.enc[r["rid"], r["m"]:r["n"] + 1]
        for c in coords[off[j]:off[j + 1]]:
            np.testing.assert_array_equal(both[c:c + len(sub)], sub)
        assert off[j + 1] - off[j] == min(int(r["s"]), 500)
