#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04f
mkdir -p $OUT
cd $R
for lib in libfemcy_hip.so libfemcy_hip_x1.so libfemcy_hip_x2.so libfemcy_hip_x3.so libfemcy_hip.so; do
  FEMCY_HIP_LIB=$R/femcy_amd/$lib ONLY6=1 timeout 200 python tools/r04_ab.py persist 2>&1 | grep -v amdgpu.ids >> $OUT/persist_micro.txt
done
cat $OUT/persist_micro.txt
timeout 900 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
