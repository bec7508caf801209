#!/bin/bash
# round 5, call 11: kernel trace of the band factorisation with the matrix-core update as the default (variant 1) on the
# 89 k-DOF cube and the dense CPS6 deck; direct / e2e tests on the new default; direct_limit at the default
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05k
mkdir -p $OUT
cd /tmp
for v in 1 0; do
  VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$v -o kt -- python $R/tools/direct_limit.py 30 > $OUT/kt_$v.log 2>&1
  echo "== FEMCY_TUNE_DIRECT_UPDATE = $v (30^3 cells, 89 k DOF, 91 tiles per panel; 6 factorisations)" >> $OUT/direct_mfma_kernels.txt
  python $R/tools/rocprof_summary.py stats $(find $OUT/kt_$v -name "*.db" | head -1) | head -6 >> $OUT/direct_mfma_kernels.txt 2>&1
  rm -rf $OUT/kt_$v
done
cat $OUT/direct_mfma_kernels.txt
cd $R
timeout 300 python tools/direct_limit.py 12 20 30 2>&1 | grep -v amdgpu.ids > $OUT/direct_limit_default.txt; cat $OUT/direct_limit_default.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_direct.py -q -m gpu > $OUT/pytest_direct.log 2>&1; tail -3 $OUT/pytest_direct.log
