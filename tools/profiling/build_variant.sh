#!/bin/bash
# A libgab_hip.so variant with extra -D flags for ONE translation unit (knock-out / tuning builds):
#   tools/profiling/build_variant.sh <name> <unit without .hip> <flags...>   ->  variants/libgab_<name>.so
# Use with GAB_LIB_PATH=variants/libgab_<name>.so.  The other objects come from genarchbench_amd/csrc/build (run make first).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; unit=$2; shift 2
cd $ROOT/genarchbench_amd/csrc
mkdir -p $ROOT/variants/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result -Wno-unused-variable "$@" -c $unit.hip -o $ROOT/variants/obj/${unit}_$name.o
objs=""
for o in build/*.o; do b=$(basename $o .o); if [ "$b" = "$unit" ]; then objs="$objs $ROOT/variants/obj/${unit}_$name.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/libgab_$name.so $objs
echo variants/libgab_$name.so
