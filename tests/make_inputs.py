#!/usr/bin/env python3
"""Builds a GENARCH_BENCH_INPUTS_ROOT-like tree (the reference's genarch-inputs layout, file names as its
regression scripts expect them) from the committed golden fixtures, so that benchmarks/*/scripts/regression_small.sh
can run end to end where the original 90 GB data set is not available.

    python tests/make_inputs.py <out_dir>
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
G = os.path.join(HERE, "golden")


def make(out):
    from tools import mkindex
    from tests.util import read_fasta_codes
    cp = lambda a, b: (os.makedirs(os.path.dirname(b), exist_ok=True), shutil.copyfile(os.path.join(G, a), b))
    cp("bsw_bench.in.txt", f"{out}/bsw/small/bandedSWA_SRR7733443_100k_input.txt")
    cp("bsw_bench.expected.txt", f"{out}/bsw/small/output-reference.file")
    cp("chain_bench.in.txt", f"{out}/chain/small/in-1k.txt")
    cp("chain_bench.chain.expected.txt", f"{out}/chain/small/out-reference.txt")
    cp("chain_bench.fastchain.expected.txt", f"{out}/chain/small/out-reference-no-heuristics-32b.txt")
    cp("bpm_adv.in.txt", f"{out}/bpm/small/BPM_SRR7733443_100k_input.txt")
    cp("bpm_adv.expected.txt", f"{out}/bpm/small/output-reference.file")
    cp("wfa_adv.in.txt", f"{out}/wfa/small/WFA_SRR7733443_100k_input.txt")
    cp("wfa_adv.expected.txt", f"{out}/wfa/small/output-reference.file")
    cp("fmi_small.reads.fq", f"{out}/fmi/small/SRR7733443_1m_1.fastq")
    cp("fmi_small.expected.txt", f"{out}/fmi/small/out-reference.txt")
    mkindex.FmIndex(read_fasta_codes(os.path.join(G, "fmi_small.ref.fa"))).write(f"{out}/fmi/broad", with_bns=True)
    return out


if __name__ == "__main__":
    print(make(sys.argv[1]))
