"""Exhaustive parity at BASELINE.json's LARGE sizes, as part of `pytest -m gpu` ("full-size parity is
builder-run only").

`tests/full_parity.py` compares EVERY item of a large configuration with the oracle: the score of all 10 M bsw and bpm
pairs and of all 10 M bitpal pairs in both of its modes, score + length + every CIGAR operation of all 1 M wfa pairs
(complete and adaptive), score and parent of all 85 M anchors of the 10 000 chain / fast-chain calls, and every field of
every SMEM of 1.5 M reads of 151 bp against a 48 Mbp index in both interval-list formats (the fmi-large workload at a size
that fits here).  It runs here as ONE child process (a second GPU process beside the
test runner, inside the box's process guard) so that its 10 M-pair buffers are gone when it returns.  fmi-large itself
(211 s of oracle time for its 77 M SMEM records, plus the 256 Mbp index build) stays in the stand-alone script
(`python tests/full_parity.py`, last run kept in profiles/rNN_full_size_parity.md); bench.py checks its first 20 000 reads
per run.
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
