#!/bin/bash
# HBM traffic of every workload's kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (the guide's rule: one
# counter per pass; --pmc is never combined with a trace domain other than --kernel-trace), one bench step each.
# Run on the GPU box from the repository root; writes gpurun_out/traffic/<workload>_<counter>.txt and, at the end,
# gpurun_out/traffic/${ROUND}_hbm_traffic.json (copy it to profiles/: bench.py reads roofline.traffic from there).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${ROUND:-r03}
mkdir -p gpurun_out/traffic
for w in ${WORKLOADS:-bsw chain fast-chain bpm bitpal bitpal-edit wfa fmi fmi-sa parse-bsw}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== $w $c" >> gpurun_out/traffic/progress.log
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/traffic/${w}_$c -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-host-roi --no-check > gpurun_out/traffic/${w}_$c.json 2> gpurun_out/traffic/${w}_$c.err || exit 1
    python3 tools/profiling/pmc_sum.py gpurun_out/traffic/${w}_$c > gpurun_out/traffic/${w}_$c.txt
  done
done
python3 tools/profiling/traffic_json.py gpurun_out/traffic > gpurun_out/traffic/${R}_hbm_traffic.json
