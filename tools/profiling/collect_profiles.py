"""Copy the summaries of tools/profiling/profile_all.sh from gpurun_out/<round>/ into profiles/ under the round's prefix:
<round>_<workload>_large_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <round>_<workload>_large_kernel_trace.csv for
the workloads whose launches overlap (bsw, bpm), and <round>_<workload>_large_bench.json (the un-profiled bench line) -- and
REFUSE a set that is not evidence for the bench line (stale chain profiles of a kernel that no longer existed):

  * every kernel bench.py prices (TRAFFIC_KERNELS[workload]) and every kernel of <round>_hbm_traffic.json's entry for the
    workload must appear in the workload's kernel_stats.csv;
  * for the dominant kernel(s), average duration x launches per step (calls / profiled steps) must not exceed the bench
    line's ms_per_step by more than 10 % (a profile of a slower, older kernel cannot pass; kernels that overlap on two
    streams -- bsw, bpm -- are checked through their first-start-to-last-end span in the trace instead).

    python tools/profiling/collect_profiles.py r03 [workload ...]        exit code 1 = refused, nothing copied for that workload"""
import csv, glob, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench  # noqa: E402  (TRAFFIC_KERNELS: the kernels the bench line prices)

PROFILED_STEPS = 4          # profile_all.sh: --steps 3 --warmup 1
OVERLAPPED = ("bsw", "bpm")


def short(name):
    """'void (anonymous namespace)::k<3, false>(args...)' -> 'k<3, false>' (the form bench.py and the traffic table use)"""
    name = name.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return name[:i]
    return name


def check(w, stats_csv, trace_csv, bench_line, traffic):
    """-> list of reasons to refuse (empty = accepted)"""
    why = []
    rows = list(csv.DictReader(open(stats_csv)))
    for r in rows:
        r["Name"] = short(r["Name"])
    names = [r["Name"] for r in rows]
    want = list(bench.TRAFFIC_KERNELS.get(w, ([], False))[0])
    if want and not any(any(n.startswith(k) for n in names) for k in want):
        why.append(f"none of the kernels bench.py prices for {w} ({want}) is in {os.path.basename(stats_csv)}")
    for k in (traffic or {}).get(w, {}):
        if k not in names:
            why.append(f"kernel {k!r} of the HBM-traffic table is not in the kernel stats")
    ms_step = bench_line.get("ms_per_step")
    if ms_step and want:
        if w in OVERLAPPED and trace_csv:
            from kernel_span import spans            # launches closer than 0.5 ms belong to one step; span = first start .. last end
            sp = sorted(x[1] for k in want for x in spans(trace_csv, k))
            if sp and sp[len(sp) // 2] > 1.10 * ms_step:
                why.append(f"span of {want} in the trace ({sp[len(sp) // 2]:.2f} ms per step) exceeds the bench line's ms_per_step {ms_step:.2f} by > 10 %")
        else:
            per_step = 0.0
            for r in rows:
                if any(r["Name"].startswith(k) for k in want):
                    per_step += float(r["AverageNs"]) * 1e-6 * int(r["Calls"]) / PROFILED_STEPS
            if per_step > 1.10 * ms_step:
                why.append(f"{want}: average x launches per step = {per_step:.2f} ms exceeds the bench line's ms_per_step {ms_step:.2f} by > 10 %")
    return why


def main():
    rnd = sys.argv[1]
    src = os.path.join(root, "gpurun_out", rnd)
    which = sys.argv[2:] or sorted({f[len("bench_"):-len(".json")] for f in os.listdir(src) if f.startswith("bench_") and f.endswith(".json")})
    tpath = os.path.join(root, "profiles", f"{rnd}_hbm_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else None
    refused = 0
    for w in which:
        # gpurun merges every call's files into the same directory: take the NEWEST collection, not the first the glob finds
        newest = lambda pat: sorted(glob.glob(os.path.join(src, f"prof_{w}", "**", pat), recursive=True), key=os.path.getmtime, reverse=True)
        stats, trace = newest("*kernel_stats.csv"), newest("*kernel_trace.csv")
        b = os.path.join(src, f"bench_{w}.json")
        line = json.loads(open(b).read().strip().splitlines()[-1]) if os.path.exists(b) else {}
        if stats:
            why = check(w, stats[0], trace[0] if trace else None, line, traffic)
            if why:
                refused += 1
                print(f"{w}: REFUSED --", "; ".join(why))
                continue
            shutil.copyfile(stats[0], os.path.join(root, "profiles", f"{rnd}_{w}_large_kernel_stats.csv"))
        if trace and w in OVERLAPPED:
            shutil.copyfile(trace[0], os.path.join(root, "profiles", f"{rnd}_{w}_large_kernel_trace.csv"))
        if line:
            open(os.path.join(root, "profiles", f"{rnd}_{w}_large_bench.json"), "w").write(json.dumps(line) + "\n")
        print(w, "stats" if stats else "-", "bench" if line else "-")
    return 1 if refused else 0


if __name__ == "__main__":
    sys.exit(main())
