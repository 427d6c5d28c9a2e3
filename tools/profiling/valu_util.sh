#!/bin/bash
# VALU issue / lane utilisation / occupancy / wait share of every workload's kernels; run on the GPU box from the repo root.
# The SQ block collects at most a handful of counters per pass (asking for more aborts rocprofv3 with "Request exceeds the
# capabilities of the hardware to collect", r01's pmcg3.log), so the list is split into passes of FOUR counters, each its
# own run of the workload; --pmc is only ever combined with --kernel-trace.
# ITEMS=1000 TAG=_1000 WORKLOADS="chain fast-chain" ...: another input size (--items), results under <workload><TAG>_p<pass>
# SHARD=0/8 TAG=_shard ...: one rank's share of an N-GPU strong-scaling run (--shard)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/valu
EXTRA=""; if [ -n "$ITEMS" ]; then EXTRA="--items $ITEMS"; fi
if [ -n "$SHARD" ]; then EXTRA="$EXTRA --shard $SHARD"; fi
PASSES=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS")
for w in ${WORKLOADS:-bsw chain fast-chain bpm wfa fmi fmi-sa parse-bsw}; do
  p=0
  for counters in "${PASSES[@]}"; do
    n=$(echo $counters | wc -w)
    if [ "$n" -gt 4 ]; then echo "a pass may hold at most 4 SQ counters" >&2; exit 1; fi
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d gpurun_out/valu/${w}${TAG}_p$p -- python3 bench.py --workload $w $EXTRA --steps 1 --warmup 0 --no-cpu-baseline --no-host-roi --no-check > gpurun_out/valu/${w}${TAG}_p$p.json 2> gpurun_out/valu/${w}${TAG}_p$p.err || exit 1
    python3 tools/profiling/pmc_sum.py gpurun_out/valu/${w}${TAG}_p$p > gpurun_out/valu/${w}${TAG}_p$p.txt
    p=$((p + 1))
  done
  echo "== $w" >> gpurun_out/valu/progress.log
done
