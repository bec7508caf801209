#!/bin/bash
# round 4, direct solve (register-resident kernels): tests, timing against the tight PCG, kernel trace
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04direct
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_direct.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "direct or gen_beam or twist_plate_C3D4 or nu0d4999 or ellip_dense" > $OUT/pytest2.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest2.log
grep -v "Warn\|^$" $OUT/pytest2.log | tail -15
timeout 600 python tools/direct_bench.py > $OUT/direct_bench.txt 2>&1
grep -v "Warn\|amdgpu.ids" $OUT/direct_bench.txt
cd /tmp && export TMPDIR=/tmp
REPS=3 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o direct -- python $R/tools/direct_bench.py ellip_dense_CPS6_0d04 twist_plate_C3D10 > $OUT/prof_run.txt 2>&1
python $R/tools/rocprof_summary.py stats $(find $OUT/prof -name "*.db" | head -1) > $OUT/direct_kernel_stats.txt
head -12 $OUT/direct_kernel_stats.txt
rm -rf $OUT/prof
