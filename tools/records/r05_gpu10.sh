#!/bin/bash
# round 5, call 10: the trailing update of the band factorisation on the f64 matrix cores (FEMCY_TUNE_DIRECT_UPDATE 1: one tile
# pair per workgroup, 2: 2 x 2 tile pairs) against the VALU product (0): correctness, time per solve, kernel times
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05j
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_direct.py -q -m gpu -k "matrix_cores" > $OUT/pytest_mfma.log 2>&1; tail -6 $OUT/pytest_mfma.log
for v in 0 1 2; do
  echo "== FEMCY_TUNE_DIRECT_UPDATE = $v" >> $OUT/direct_mfma.txt
  (VARIANT=$v timeout 300 python tools/direct_limit.py 12 20 30; VARIANT=$v REPS=5 timeout 200 python tools/direct_bench.py twist_plate_C3D10 ellip_dense_CPS6_0d04) 2>&1 | grep -v amdgpu.ids >> $OUT/direct_mfma.txt
done
cat $OUT/direct_mfma.txt
cd /tmp
for v in 0 2; do
  VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$v -o kt -- python $R/tools/direct_limit.py 30 > $OUT/kt_$v.log 2>&1
  echo "== FEMCY_TUNE_DIRECT_UPDATE = $v (30^3 cells, 89 k DOF, 91 tiles per panel)" >> $OUT/direct_mfma_kernels.txt
  python $R/tools/rocprof_summary.py stats $(find $OUT/kt_$v -name "*.db" | head -1) | head -8 >> $OUT/direct_mfma_kernels.txt 2>&1
  rm -rf $OUT/kt_$v
done
cat $OUT/direct_mfma_kernels.txt
