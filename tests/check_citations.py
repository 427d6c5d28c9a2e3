#!/usr/bin/env python3
"""Every `path/file.ext:line[-line]` citation of the reference in the headers, the oracle, the library sources and the documents:
does the file exist under the reference's benchmarks/ tree and does it have that many lines?  (Build container only: the reference
is not on the GPU box.)    python tests/check_citations.py [/root/reference/benchmarks]"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/benchmarks"
CITE = re.compile(r"((?:bsw|bpm|wfa|chain|fast-chain|fmi)/[A-Za-z0-9_./+-]+\.(?:cpp|c|h|sh|py)):(\d+)(?:-(\d+))?")
BARE = re.compile(r"(?<![A-Za-z0-9_./-])([A-Za-z][A-Za-z0-9_]*\.(?:cpp|c|h)):(\d+)(?:-(\d+))?")      # "bandedSWA.cpp:48-66": any file of that name
OURS = {"gab.h", "gab_internal.h", "chain_dev.h", "gab_driver.h", "oracle.h", "main.c", "main_banded.c", "fmi.c", "align_benchmark.c"}


def main():
    if not os.path.isdir(REF):
        print(f"{REF}: not here -- nothing checked")
        return 0
    files = [os.path.join(ROOT, f) for f in ("include/gab.h", "INTEGRATION.md", "DESIGN.md", "README.md", "SURVEY.md")]
    for pat in ("oracle/*.c", "oracle/*.h", "oracle/*.py", "genarchbench_amd/csrc/*.hip", "genarchbench_amd/csrc/*.h", "genarchbench_amd/*.py",
                "benchmarks/*/src/*.c", "benchmarks/*/tools/*.c", "benchmarks/*/*.c", "benchmarks/common/*.h", "tests/*.py", "bench.py"):
        files += sorted(glob.glob(os.path.join(ROOT, pat)))
    lines_of = {}
    bad = total = 0
    for f in files:
        if not os.path.exists(f) or os.path.basename(f) == "SURVEY.md":
            continue
        for ln, text in enumerate(open(f, errors="replace"), 1):
            for m in CITE.finditer(text):
                path, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
                total += 1
                full = os.path.join(REF, path)
                if full not in lines_of:
                    lines_of[full] = sum(1 for _ in open(full, errors="replace")) if os.path.isfile(full) else -1
                n = lines_of[full]
                if n < 0:
                    print(f"{os.path.relpath(f, ROOT)}:{ln}: {path}: no such file in the reference"); bad += 1
                elif b > n or a > b:
                    print(f"{os.path.relpath(f, ROOT)}:{ln}: {path}:{a}-{b}: the file has {n} lines"); bad += 1
    by_name = {}
    for dp, _, fs in os.walk(REF):
        for fn in fs:
            by_name.setdefault(fn, []).append(os.path.join(dp, fn))
    for f in files:
        if not os.path.exists(f) or os.path.basename(f) == "SURVEY.md":
            continue
        for ln, text in enumerate(open(f, errors="replace"), 1):
            for m in BARE.finditer(text):
                name, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
                if name not in by_name:
                    if name not in OURS and not os.path.exists(os.path.join(ROOT, "genarchbench_amd", "csrc", name)):
                        print(f"{os.path.relpath(f, ROOT)}:{ln}: {name}: no file of that name in the reference"); bad += 1; total += 1
                    continue
                if name in OURS and f.endswith((".md",)) and "benchmarks/" in text:
                    continue
                total += 1
                longest = max(sum(1 for _ in open(c, errors="replace")) for c in by_name[name])
                if b > longest or a > b:
                    print(f"{os.path.relpath(f, ROOT)}:{ln}: {name}:{a}-{b}: the longest file of that name has {longest} lines"); bad += 1
    print(f"{total} citations checked, {bad} do not resolve")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
