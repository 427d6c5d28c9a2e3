#!/bin/bash
# rocprofv3 kernel stats + bench JSON for every workload (run on the GPU box from the repo root).
#   ROUND=r03 WORKLOADS="bsw chain" bash tools/profiling/profile_all.sh  -> gpurun_out/$ROUND/{prof_<w>/, prof_<w>.json, bench_<w>.json}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${ROUND:-r03}
mkdir -p gpurun_out/$R
for w in ${WORKLOADS:-bsw chain fast-chain bpm bitpal bitpal-edit wfa fmi fmi-sa parse-bsw}; do
  echo "== $w" >> gpurun_out/$R/progress.log
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof_$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-host-roi > gpurun_out/$R/prof_$w.json 2> gpurun_out/$R/prof_$w.err || exit 1
  python3 bench.py --workload $w --steps 5 --warmup 2 > gpurun_out/$R/bench_$w.json 2> gpurun_out/$R/bench_$w.err || exit 1
  tail -c 600 gpurun_out/$R/bench_$w.json >> gpurun_out/$R/progress.log
done
