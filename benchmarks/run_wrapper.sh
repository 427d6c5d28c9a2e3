#!/bin/bash
#
# Local job runner for the regression scripts (source it, like the reference's run_wrapper.sh).
# Same contract as /root/reference/benchmarks/run_wrapper.sh:8-99 for the variables the regression scripts set
# (job, clean, commands[], parallelism[], command_opts, before_command, before_run(), after_run()), with one
# extra dimension in each parallelism entry: gpus=G (exported as GAB_GPUS).  Batch schedulers (SLURM / PJM)
# are out of scope: a GPU node is driven directly.
#
#    parallelism=( 'nodes=1, mpi=1, omp=1, gpus=1' 'nodes=1, mpi=1, omp=1, gpus=8' )
#
if test -t 1 && test -n "$(tput colors 2>/dev/null)" && test "$(tput colors)" -ge 8; then
    RED='\033[01;31m'; GREEN='\033[01;32m'; COLOR_RESTORE='\033[0m'
fi
: "${clean:=1}" "${job:=JOB}"
gab_wrapper_failed=0
gab_stage_root="$(pwd)"
for gab_cmd in "${commands[@]}"; do
    for gab_par in "${parallelism[@]}"; do
        omp=1; gpus=1; nodes=1; mpi=1
        for kv in ${gab_par//,/ }; do
            case "$kv" in omp=*) omp="${kv#omp=}";; gpus=*) gpus="${kv#gpus=}";; nodes=*) nodes="${kv#nodes=}";; mpi=*) mpi="${kv#mpi=}";; esac
        done
        gab_name="${job}_$(basename "${gab_cmd%% *}")_nodes_${nodes}_mpi_${mpi}_omp_${omp}_gpus_${gpus}_$(date +%Y%m%d_%H%M%S)"
        mkdir -p "$gab_stage_root/$gab_name" && cd "$gab_stage_root/$gab_name" || exit 1
        before_run "$gab_name"
        export OMP_NUM_THREADS="$omp" MPI_RANKS="$mpi" GAB_GPUS="$gpus"
        printf '#!/bin/bash\nexport OMP_NUM_THREADS=%s MPI_RANKS=%s GAB_GPUS=%s\n%s %s %s\n' "$omp" "$mpi" "$gpus" \
               "$before_command" "$gab_cmd" "$command_opts" > "$gab_name.sh"
        bash "$gab_name.sh" 1> "$gab_name.out" 2> "$gab_name.err"
        gab_rc=$?
        gab_msg=""
        if [[ $gab_rc -eq 0 ]]; then gab_msg="$(after_run "$gab_name")"; gab_rc=$?; else gab_msg="command exited with $gab_rc"; fi
        if [[ $gab_rc -eq 0 ]]; then
            echo -e "${GREEN}OK${COLOR_RESTORE}      $gab_name  $gab_msg"
            cd "$gab_stage_root" && [[ "$clean" == "1" ]] && rm -rf "$gab_stage_root/$gab_name"
        else
            echo -e "${RED}FAILED${COLOR_RESTORE}  $gab_name  $gab_msg"
            gab_wrapper_failed=1
            cd "$gab_stage_root"
        fi
    done
done
[[ $gab_wrapper_failed -eq 0 ]]
