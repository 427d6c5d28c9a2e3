#!/usr/bin/env python3
"""Union span of a kernel's launches per bench step, from a rocprofv3 --kernel-trace CSV.

bsw launches one DP kernel per query-length class and runs them two at a time (two streams), so the per-launch durations
in the --stats summary overlap and their sum is about twice the time the GPU spends; the figure that matches bench.py's
event-timed `dominant_kernel_ms` is the span from the first launch's start to the last launch's end within a step.

usage: kernel_span.py <kernel_trace.csv> <kernel name substring> [gap_ms between steps, default 0.5]
"""
import csv
import sys


def spans(path, name, gap_ms=0.5):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(path)) if name in r["Kernel_Name"])
    groups = []
    for s, e in iv:
        if groups and s - max(x[1] for x in groups[-1]) <= gap_ms * 1e6:
            groups[-1].append((s, e))
        else:
            groups.append([(s, e)])
    return [(len(g), (max(x[1] for x in g) - min(x[0] for x in g)) / 1e6, sum(e - s for s, e in g) / 1e6) for g in groups]


if __name__ == "__main__":
    gap = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    for n, span, total in spans(sys.argv[1], sys.argv[2], gap):
        print(f"{n} launches: span {span:.3f} ms, sum of durations {total:.3f} ms")
