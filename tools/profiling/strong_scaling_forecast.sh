#!/bin/bash
# What the sharding of `bench.py --gpus N` (strong scaling: ONE large input split across the ranks) can reach, measured on
# ONE GPU: every rank's share of an N-GPU run is processed on GPU 0 in turn (bench.py --shard R/N); the N-GPU step time is the
# slowest rank's, the forecast throughput T / that.  Run on the GPU box from the repo root:
#   WORKLOADS="bsw chain fast-chain" NS="2 4 8" bash tools/profiling/strong_scaling_forecast.sh > gpurun_out/forecast.md
cd $GRAFT_REPO_ROOT
echo "| workload | N | slowest rank's step (ms) | fastest rank's (ms) | forecast, whole job | speed-up over N = 1 | efficiency |"
echo "|---|---|---|---|---|---|---|"
for w in ${WORKLOADS:-bsw chain fast-chain bpm wfa}; do
  base=$(python3 bench.py --workload $w --steps 5 --no-cpu-baseline --no-host-roi --no-check 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['total_items'], d['value'], d['unit'].replace(' ','_'))")
  set -- $base; ms1=$1; T=$2; v1=$3; unit=$4
  echo "| $w-large | 1 | $ms1 | $ms1 | $v1 ${unit//_/ } | 1.00 | 100 % |"
  for n in ${NS:-2 4 8}; do
    worst=0; best=1e9; units=0
    for ((r = 0; r < n; r++)); do
      line=$(python3 bench.py --workload $w --shard $r/$n --steps 5 --no-cpu-baseline --no-host-roi --no-check 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'] * d['ms_per_step'] * 1e3)")
      set -- $line
      worst=$(python3 -c "print(max($worst, $1))"); best=$(python3 -c "print(min($best, $1))"); units=$(python3 -c "print($units + $2)")
    done
    python3 -c "v = $units / ($worst * 1e-3) / 1e6; print(f'| $w-large | $n | {$worst:.3f} | {$best:.3f} | {v:.1f} ${unit//_/ } | {v / $v1:.2f} | {100 * v / $v1 / $n:.0f} % |')"
  done
done
