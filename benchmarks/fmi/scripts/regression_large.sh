#!/bin/bash
# FMI regression on the large input: runs the MI355X driver at 1 GPU (and at $GAB_REGRESSION_GPUS if set) and
# diffs its output with the expected file of the data set, exactly like the reference's script of the same name.
inputs_path="$GENARCH_BENCH_INPUTS_ROOT/fmi/large"
if [[ -z "$GENARCH_BENCH_INPUTS_ROOT" || ! -d "$inputs_path" ]]; then
    echo "ERROR: You have not set a valid input folder $inputs_path"
    exit 1
fi
scriptfolder="$(dirname "$(realpath "$0")")"
binaries_path="$(dirname "$scriptfolder")"
clean=1
job="FMI-REGRESSION-LARGE"
before_command=""
# $GAB_FMI_COMMAND substitutes another binary with the same CLI (e.g. the compiled reference, to run this harness on a box without a GPU)
commands=( "${GAB_FMI_COMMAND:-$binaries_path/fmi}" )
parallelism=( 'nodes=1, mpi=1, omp=1, gpus=1' )
[[ -n "$GAB_REGRESSION_GPUS" ]] && parallelism+=( "nodes=1, mpi=1, omp=1, gpus=$GAB_REGRESSION_GPUS" )
command_opts="\"$GENARCH_BENCH_INPUTS_ROOT/fmi/broad\" \"$inputs_path/SRR7733443_10m_1.fastq\" ${GAB_FMI_BATCH:-512} 19 \$OMP_NUM_THREADS"
before_run() ( job_name="$1" )
after_run() (
    job_name="$1"
    kernel_time="$(sed -n 6p "$job_name.out" | cut -d " " -f 3)"
    sed -n "7~1p" "$job_name.out" | diff --brief - "$inputs_path/out-reference.txt" >/dev/null 2>&1 || { echo "The output file is not identical to the reference file"; return 1; }
    echo "Kernel execution time $kernel_time s"
    grep "Energy consumption:" "$job_name.err"
    return 0
)
source "$scriptfolder/../../run_wrapper.sh"
