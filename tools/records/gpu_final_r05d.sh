#!/bin/bash
# round-5 record run, closing: the whole GPU suite plain and under FEMCY_DEBUG_POISON=1 on the final sources, smoke
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05final
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
FEMCY_DEBUG_POISON=1 timeout 1500 python -m pytest tests/ -q -m gpu -p no:faulthandler > $OUT/pytest_gpu_poison.log 2>&1; tail -3 $OUT/pytest_gpu_poison.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
