#!/bin/bash
# end-to-end wall time of the bsw driver on the large input, line-by-line parser vs GAB_GPU_PARSE=1 (run on the GPU box)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from tools import gabgen
gabgen.write_text("bsw", "/tmp/bsw_large.txt", 2, 10_000_000, 0)
print("written", os.path.getsize("/tmp/bsw_large.txt"))
PY
for mode in 0 1; do
  t0=$(date +%s.%N)
  GAB_GPU_PARSE=$mode GAB_GPUS=1 ./benchmarks/bsw/main_bsw -pairs /tmp/bsw_large.txt -t 1 -b 512 2> /tmp/err_$mode.txt | grep -E "Read time|Overall SW|parsed"
  t1=$(date +%s.%N)
  echo "GAB_GPU_PARSE=$mode wall $(python3 -c "print(round($t1 - $t0, 2))") s; last score line: $(tail -1 /tmp/err_$mode.txt); md5 of all scores $(md5sum /tmp/err_$mode.txt | cut -c1-12)"
done
