#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04e
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_pcg_persist.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_a.log 2>&1; tail -8 $OUT/pytest_a.log
for wl in c3d10 c3d4 c3d4_8m; do timeout 400 python tools/r04_ab.py fused $wl 2>&1 | grep -v amdgpu.ids >> $OUT/ab_fused.txt; done
cat $OUT/ab_fused.txt
timeout 900 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log
