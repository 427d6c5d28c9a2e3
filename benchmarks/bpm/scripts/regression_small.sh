#!/bin/bash
# BPM regression on the small input: runs the MI355X driver at 1 GPU (and at $GAB_REGRESSION_GPUS if set) and
# diffs its output with the expected file of the data set, exactly like the reference's script of the same name.
inputs_path="$GENARCH_BENCH_INPUTS_ROOT/bpm/small"
if [[ -z "$GENARCH_BENCH_INPUTS_ROOT" || ! -d "$inputs_path" ]]; then
    echo "ERROR: You have not set a valid input folder $inputs_path"
    exit 1
fi
scriptfolder="$(dirname "$(realpath "$0")")"
binaries_path="$(dirname "$scriptfolder")"
clean=1
job="BPM-REGRESSION-SMALL"
before_command=""
# $GAB_BPM_COMMAND substitutes another binary with the same CLI (e.g. the compiled reference, to run this harness on a box without a GPU)
commands=( "${GAB_BPM_COMMAND:-$binaries_path/bin/align_benchmark}" )
parallelism=( 'nodes=1, mpi=1, omp=1, gpus=1' )
[[ -n "$GAB_REGRESSION_GPUS" ]] && parallelism+=( "nodes=1, mpi=1, omp=1, gpus=$GAB_REGRESSION_GPUS" )
command_opts="-a bpm-edit -i \"$inputs_path/BPM_SRR7733443_100k_input.txt\" -o checksum.file -t \$OMP_NUM_THREADS"
before_run() ( job_name="$1" )
after_run() (
    job_name="$1"
    kernel_time="$(grep "Time.Benchmark" "$job_name.err" | tr -s " " | cut -d " " -f 3,4)"
    sort -n -t "[" -k 2,2 checksum.file | diff --brief - "$inputs_path/output-reference.file" >/dev/null 2>&1 || { echo "The output file is not identical to the reference file"; return 1; }
    echo "Kernel execution time $kernel_time"
    grep "Energy consumption:" "$job_name.err"
    return 0
)
source "$scriptfolder/../../run_wrapper.sh"
