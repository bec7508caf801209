#!/bin/bash
# round 5, call 14: the matrix-core update without LDS (FEMCY_TUNE_DIRECT_UPDATE = 4) against variant 1
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05n
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_direct.py -q -m gpu -k "matrix_cores" > $OUT/pytest_mfma.log 2>&1; tail -4 $OUT/pytest_mfma.log
for v in 1 4; do
  echo "== FEMCY_TUNE_DIRECT_UPDATE = $v" >> $OUT/direct_nolds.txt
  (VARIANT=$v timeout 300 python tools/direct_limit.py 12 20 30; VARIANT=$v REPS=5 timeout 200 python tools/direct_bench.py twist_plate_C3D10 ellip_dense_CPS6_0d04) 2>&1 | grep -v amdgpu.ids >> $OUT/direct_nolds.txt
done
cat $OUT/direct_nolds.txt
