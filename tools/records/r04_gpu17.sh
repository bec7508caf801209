#!/bin/bash
# round 4: kernel trace of the shipped direct-solve kernels
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04direct
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPS=3 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o direct -- python $R/tools/direct_bench.py ellip_dense_CPS6_0d04 twist_plate_C3D10 > $OUT/prof_run.txt 2>&1
python $R/tools/rocprof_summary.py stats $(find $OUT/prof -name "*.db" | head -1) > $OUT/direct_kernel_stats.txt
grep "kernel  \|k_band" $OUT/direct_kernel_stats.txt | head -12
rm -rf $OUT/prof
