"""helpers shared by the tests: golden-vector readers for the reference's text formats"""
import os
import re

import numpy as np

from tools import gabgen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_bsw_input(path):
    """reference bsw input format (bsw/src/main_banded.cpp:152-206): h0 / ref digits / query digits"""
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    n = len(lines) // 3
    refs, qrys, h0s = [], [], []
    for i in range(n):
        h0s.append(int(lines[3 * i]))
        refs.append(np.frombuffer(lines[3 * i + 1], np.uint8) - 48)
        qrys.append(np.frombuffer(lines[3 * i + 2], np.uint8) - 48)
    return gabgen.bsw_from_arrays(refs, qrys, h0s)


def read_scores(path):
    """'[i] score=s' lines -> int32 array indexed by i"""
    out = {}
    for line in open(path):
        m = re.match(r"\[(\d+)\] score=(-?\d+)", line)
        if m:
            out[int(m.group(1))] = int(m.group(2))
    return np.array([out[i] for i in range(len(out))], np.int32)


def read_bsw_full(path):
    """'[i] score qle tle gtle gscore max_off' lines (oracle/ref_harness/bsw_full_ref.cpp) -> int32 [n, 6]"""
    rows = [[int(v) for v in line.split()[1:]] for line in open(path) if line.startswith("[")]
    return np.array(rows, np.int32).reshape(-1, 6)


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def read_chain_output(path):
    """reference chain output (chain/src/host_data_io.cpp:53-60): n / score<TAB>parent x n / EOR"""
    sc, pa = [], []
    for line in open(path):
        f = line.split()
        if len(f) == 2:
            sc.append(int(f[0])); pa.append(int(f[1]))
    return np.array(sc, np.int32), np.array(pa, np.int32)


def read_cigars(path):
    """'id=N CIGAR' lines -> list indexed by id"""
    out = {}
    for line in open(path):
        m = re.match(r"id=(\d+) (\S*)", line)
        if m:
            out[int(m.group(1))] = m.group(2)
    return [out[i] for i in range(len(out))]


def read_fasta_codes(path):
    seq = b"".join(l.strip() for l in open(path, "rb") if not l.startswith(b">"))
    lut = np.full(256, 4, np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    return lut[np.frombuffer(seq, np.uint8)]


def read_fastq_reads(path):
    """FASTQ -> ReadBatch the way fmi.cpp:121-151 encodes it (row stride = longest read, A C G T -> 0..3, else 4)"""
    lines = open(path, "rb").read().split(b"\n")
    seqs = [lines[i] for i in range(1, len(lines), 4) if i < len(lines) and lines[i - 1].startswith(b"@")]
    lut = np.full(256, 4, np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    stride = max(len(s) for s in seqs)
    enc = np.full((len(seqs), stride), 4, np.uint8)
    for r, s in enumerate(seqs):
        enc[r, :len(s)] = lut[np.frombuffer(s, np.uint8)]
    return gabgen.ReadBatch(enc, np.array([len(s) for s in seqs], np.int32))
