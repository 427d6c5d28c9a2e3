#!/bin/bash
# The drop-in drivers' own region of interest on the LARGE inputs (what a user of the reference's harness sees: host
# pointers in, host pointers out, PCIe both ways), next to the compiled reference on the box's host cores when oracle/_ref
# travelled.  Run on the GPU box:  bash tests/driver_roi.sh [workers ...]  -> one line per run on stdout
cd "$GRAFT_REPO_ROOT" || exit 1
T=/tmp/gab_roi; mkdir -p $T
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from tools import gabgen
T = "/tmp/gab_roi"
if not os.path.exists(f"{T}/bsw.txt"):
    gabgen.write_text("bsw", f"{T}/bsw.txt", 2, 10_000_000, 0)
if not os.path.exists(f"{T}/chain.txt"):
    gabgen.write_text("chain", f"{T}/chain.txt", 5, 10_000, 0, 50, 60000)
if not os.path.exists(f"{T}/bpm.txt"):
    gabgen.write_text("bpm", f"{T}/bpm.txt", 3, 10_000_000, 0, 151)
if not os.path.exists(f"{T}/wfa.txt"):
    gabgen.write_text("wfa", f"{T}/wfa.txt", 4, 1_000_000, 0, 151)
PY
cores=$(python3 -c "
import os
n = len(os.sched_getaffinity(0))
try:
    q, p = open(\"/sys/fs/cgroup/cpu.max\").read().split()
    n = n if q == \"max\" else max(1, min(n, int(q) // int(p)))
except Exception:
    pass
print(n)")   # the CPU quota of the box, not every visible hardware thread
if [[ -n "$GAB_ROI_FMI" ]]; then     # fmi-large: 256 Mbp index + 10 M reads of 151 bp (several minutes: index build on the host)
  python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from tools import gabgen, mkindex
T = "/tmp/gab_roi"
if not os.path.exists(f"{T}/fmi_reads.fq"):
    ref = gabgen.fmi_ref(7, 256_000_000, 5)
    mkindex.FmIndex(ref).write(f"{T}/fmi_ref", with_bns=True)
    gabgen.fmi_write_fastq(f"{T}/fmi_reads.fq", gabgen.fmi_reads(8, ref, 10_000_000, 151, 151))
PY
  GAB_WORKERS_PER_GPU=3 GAB_GPUS=1 ./benchmarks/fmi/fmi $T/fmi_ref $T/fmi_reads.fq 512 19 1 > $T/fmi_out.txt 2> $T/fmi_err.txt
  grep -E "Computing time|totalSmems" $T/fmi_out.txt | sed "s/^/fmi-large driver, 3 worker(s) per GPU: /"
  echo "   md5 of the SMEM lines: $(tail -n +7 $T/fmi_out.txt | md5sum | cut -c1-12)"
  if [[ -x oracle/_ref/fmi_ref ]]; then
    OMP_PROC_BIND=true OMP_PLACES=cores oracle/_ref/fmi_ref $T/fmi_ref $T/fmi_reads.fq 512 19 $cores > $T/fmi_ref_out.txt 2> $T/fmi_ref_err.txt
    grep -E "Computing time|totalSmems" $T/fmi_ref_out.txt | sed "s/^/fmi-large reference, $cores threads: /"
    echo "   md5 of the SMEM lines: $(tail -n +7 $T/fmi_ref_out.txt | md5sum | cut -c1-12)"
  fi
  exit 0
fi
for w in ${@:-1 2 3}; do   # chain-large through the driver needs its fscanf parse of 85 M anchors (~20 s) per run
  export GAB_WORKERS_PER_GPU=$w GAB_GPUS=1 GAB_GPU_PARSE=0      # the host-pointer path: line readers, page-locked slabs, chunks x workers
  ./benchmarks/bsw/main_bsw -pairs $T/bsw.txt -t 1 -b 512 2> $T/bsw_err_$w.txt | grep -E "Overall SW" | sed "s/^/bsw-large driver, $w worker(s) per GPU: /"
  echo "   md5 of the scores: $(grep score= $T/bsw_err_$w.txt | md5sum | cut -c1-12)"
  ./benchmarks/chain/chain -i $T/chain.txt -o $T/chain_out_$w.txt -t 1 2>&1 | grep -E "Time in kernel|region of interest" | sed "s/^/chain-large driver, $w worker(s) per GPU: /"
  echo "   md5 of the output: $(md5sum $T/chain_out_$w.txt | cut -c1-12)"
  ./benchmarks/fast-chain/chain -i $T/chain.txt -o $T/fchain_out_$w.txt -t 1 2>&1 | grep -E "Time in kernel|region of interest" | sed "s/^/fast-chain-large driver, $w worker(s) per GPU: /"
  ./benchmarks/bpm/bin/align_benchmark -a bpm-edit -i $T/bpm.txt -o $T/bpm_out_$w.txt -t 1 2>&1 | grep "Time.Benchmark" | tr -s " " | sed "s/^/bpm-large driver, $w worker(s) per GPU: /"
  echo "   md5 of the sorted output: $(sort -n -t "[" -k 2,2 $T/bpm_out_$w.txt | md5sum | cut -c1-12)"
  ./benchmarks/wfa/bin/align_benchmark -i $T/wfa.txt -o $T/wfa_out_$w.txt -t 1 2>&1 | grep "Time.Alignment" | sed "s/^/wfa-large driver, $w worker(s) per GPU: /"
  echo "   md5 of the sorted output: $(sort -n -t "=" -k 2,2 $T/wfa_out_$w.txt | md5sum | cut -c1-12)"
done
# the same drivers with the f1 GPU parsers (GAB_GPU_PARSE=1): the read phase leaves the file on the GPU, the region of interest is
# kernels + results back
if [[ -z "$GAB_ROI_NO_GPU_PARSE" ]]; then
  export GAB_WORKERS_PER_GPU=1 GAB_GPUS=1 GAB_GPU_PARSE=1
  ./benchmarks/bsw/main_bsw -pairs $T/bsw.txt -t 1 -b 512 2> $T/bsw_err_gp.txt | grep -E "Overall SW" | sed "s/^/bsw-large driver, GPU parse: /"
  echo "   md5 of the scores: $(grep score= $T/bsw_err_gp.txt | md5sum | cut -c1-12)"
  ./benchmarks/chain/chain -i $T/chain.txt -o $T/chain_out_gp.txt -t 1 2>&1 | grep -E "Time in kernel|region of interest" | sed "s/^/chain-large driver, GPU parse: /"
  echo "   md5 of the output: $(md5sum $T/chain_out_gp.txt | cut -c1-12)"
  ./benchmarks/fast-chain/chain -i $T/chain.txt -o $T/fchain_out_gp.txt -t 1 2>&1 | grep -E "Time in kernel|region of interest" | sed "s/^/fast-chain-large driver, GPU parse: /"
  ./benchmarks/bpm/bin/align_benchmark -a bpm-edit -i $T/bpm.txt -o $T/bpm_out_gp.txt -t 1 2>&1 | grep "Time.Benchmark" | tr -s " " | sed "s/^/bpm-large driver, GPU parse: /"
  echo "   md5 of the sorted output: $(sort -n -t "[" -k 2,2 $T/bpm_out_gp.txt | md5sum | cut -c1-12)"
  ./benchmarks/wfa/bin/align_benchmark -i $T/wfa.txt -o $T/wfa_out_gp.txt -t 1 2>&1 | grep "Time.Alignment" | sed "s/^/wfa-large driver, GPU parse: /"
  echo "   md5 of the sorted output: $(sort -n -t "=" -k 2,2 $T/wfa_out_gp.txt | md5sum | cut -c1-12)"
  unset GAB_GPU_PARSE
fi
if [[ -x oracle/_ref/bsw_ref_avx512 ]]; then
  OMP_PROC_BIND=true OMP_PLACES=cores oracle/_ref/bsw_ref_avx512 -pairs $T/bsw.txt -t $cores -b 512 2> $T/bsw_ref_err.txt | grep -E "Overall SW" | sed "s/^/bsw-large reference (avx512), $cores threads: /"
  echo "   md5 of the scores: $(grep score= $T/bsw_ref_err.txt | head -10000000 | md5sum | cut -c1-12)"
  OMP_PROC_BIND=true OMP_PLACES=cores oracle/_ref/chain_ref -i $T/chain.txt -o $T/chain_ref_out.txt -t $cores 2>&1 | grep "Time in kernel" | sed "s/^/chain-large reference, $cores threads: /"
  echo "   md5 of the output: $(md5sum $T/chain_ref_out.txt | cut -c1-12)"
  OMP_PROC_BIND=true OMP_PLACES=cores oracle/_ref/bpm_ref -a bpm-edit -i $T/bpm.txt -o $T/bpm_ref_out.txt -t $cores 2>&1 | grep "Time.Benchmark" | tr -s " " | sed "s/^/bpm-large reference, $cores threads: /"
  echo "   md5 of the sorted output: $(sort -n -t "[" -k 2,2 $T/bpm_ref_out.txt | md5sum | cut -c1-12)"
  OMP_PROC_BIND=true OMP_PLACES=cores oracle/_ref/wfa_ref -i $T/wfa.txt -o $T/wfa_ref_out.txt -t $cores 2>&1 | grep "Time.Alignment" | sed "s/^/wfa-large reference, $cores threads: /"
  echo "   md5 of the sorted output: $(sort -n -t "=" -k 2,2 $T/wfa_ref_out.txt | md5sum | cut -c1-12)"
fi
