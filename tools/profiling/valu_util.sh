#!/bin/bash
# VALU issue / lane utilisation of every workload's kernels (one PMC pass per workload); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/valu
for w in ${WORKLOADS:-bsw chain fast-chain bpm wfa fmi fmi-sa parse-bsw}; do
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/valu/$w -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-check > gpurun_out/valu/$w.json 2> gpurun_out/valu/$w.err || exit 1
  echo "== $w" >> gpurun_out/valu/progress.log
done
