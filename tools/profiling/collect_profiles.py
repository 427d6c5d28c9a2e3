"""Copy the summaries of tools/profiling/profile_all.sh from gpurun_out/<round>/ into profiles/ under the round's prefix:
<round>_<workload>_large_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <round>_<workload>_large_kernel_trace.csv for
the workloads whose launches overlap (bsw, bpm), and <round>_<workload>_large_bench.json (the un-profiled bench line).
    python tools/profiling/collect_profiles.py r02 [workload ...]"""
import glob, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rnd = sys.argv[1]
src = os.path.join(root, "gpurun_out", rnd)
which = sys.argv[2:] or sorted({f[len("bench_"):-len(".json")] for f in os.listdir(src) if f.startswith("bench_") and f.endswith(".json")})
for w in which:
    # gpurun merges every call's files into the same directory: take the NEWEST collection, not the first the glob finds
    newest = lambda pat: sorted(glob.glob(os.path.join(src, f"prof_{w}", "**", pat), recursive=True), key=os.path.getmtime, reverse=True)
    stats, trace = newest("*kernel_stats.csv"), newest("*kernel_trace.csv")
    if stats:
        shutil.copyfile(stats[0], os.path.join(root, "profiles", f"{rnd}_{w}_large_kernel_stats.csv"))
    if trace and w in ("bsw", "bpm"):
        shutil.copyfile(trace[0], os.path.join(root, "profiles", f"{rnd}_{w}_large_kernel_trace.csv"))
    b = os.path.join(src, f"bench_{w}.json")
    if os.path.exists(b):
        line = [l for l in open(b).read().splitlines() if l.startswith("{")][-1]
        open(os.path.join(root, "profiles", f"{rnd}_{w}_large_bench.json"), "w").write(line + "\n")
    print(w, "stats" if stats else "-", "bench" if os.path.exists(b) else "-")
