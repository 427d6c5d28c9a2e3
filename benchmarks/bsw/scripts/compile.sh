#!/bin/bash
# Builds libgab_hip.so (hipcc, gfx950) and this benchmark's driver.
scriptfolder="$(dirname "$(realpath "$0")")"
make -C "$scriptfolder/../.." PERF_ANALYSIS=${PERF_ANALYSIS:-0} bsw/main_bsw
