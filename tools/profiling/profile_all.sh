#!/bin/bash
# rocprofv3 kernel stats + bench JSON for every workload (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
for w in ${WORKLOADS:-bsw chain fast-chain bpm wfa fmi fmi-sa}; do
  echo "== $w" >> gpurun_out/r01/progress.log
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01/prof_$w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01/prof_$w.json 2> gpurun_out/r01/prof_$w.err || exit 1
  python3 bench.py --workload $w --steps 5 --warmup 2 > gpurun_out/r01/bench_$w.json 2> gpurun_out/r01/bench_$w.err || exit 1
  tail -c 600 gpurun_out/r01/bench_$w.json >> gpurun_out/r01/progress.log
done
