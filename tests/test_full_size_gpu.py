"""Exhaustive parity at BASELINE.json's LARGE sizes, as part of `pytest -m gpu` ("full-size parity is
builder-run only").

`tests/full_parity.py` compares EVERY item of a large configuration with the oracle: the score of all 10 M bsw and bpm
pairs and of all 10 M bitpal pairs in both of its modes, score + length + every CIGAR operation of all 1 M wfa pairs
(complete and adaptive), score and parent of all 85 M anchors of the 10 000 chain / fast-chain calls, and every field of
every SMEM of 1.5 M reads of 151 bp against a 48 Mbp index in both interval-list formats (the fmi-large workload at a size
that fits here).  It runs here as ONE child process (a second GPU process beside the
test runner, inside the box's process guard) so that its 10 M-pair buffers are gone when it returns.
r04: fmi-large ITSELF is the second test -- the 256 Mbp index, all 10 M reads of 151 bp, every field of every one
of the ~77 M SMEMs against the oracle on all host threads (~3 minutes of oracle time: fmi/fmi.cpp:288-348 is the reference loop);
bench.py still checks the first 20 000 reads of every run.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ["bsw", "bpm", "bitpal", "wfa", "chain", "fast-chain", "chain-shard", "fast-chain-shard", "fmi-mid"]


@pytest.mark.gpu
def test_every_item_of_the_large_configurations():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "full_parity.py")] + WORKLOADS, cwd=ROOT,
                       capture_output=True, text=True, timeout=1100)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("| ") and "-large" in ln]
    assert len(rows) == 12, r.stdout     # bsw, bpm, bitpal x 2, wfa x 2, chain, fast-chain, their 8-GPU shards, fmi x 2 list formats
    assert all("| identical |" in ln for ln in rows), r.stdout
    assert "ALL IDENTICAL" in r.stdout


@pytest.mark.gpu
def test_fmi_large_every_smem_of_all_10m_reads():
    """BASELINE.json configs[4]'s fmi-large at full size: 10 M reads against the 256 Mbp index (512 M BWT rows, 0.5 GB of CP_OCC),
    all six fields of every SMEM and the per-read offsets identical to the oracle's (a child process: 6 GB of SMEM records on
    either side are gone when it returns)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "full_parity.py"), "fmi"], cwd=ROOT, capture_output=True, text=True, timeout=850)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("| fmi-large (256 Mbp index)")]
    assert len(rows) == 1 and "| 10000000 reads," in rows[0] and "| identical |" in rows[0], r.stdout
