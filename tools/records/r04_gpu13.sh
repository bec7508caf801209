#!/bin/bash
# round 4, direct solve: kernel trace of the solve of two decks
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04direct
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPS=3 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o direct -- python $GRAFT_REPO_ROOT/tools/direct_bench.py ellip_dense_CPS6_0d04 twist_plate_C3D10 > $OUT/prof_run.txt 2>&1
grep -v Warn $OUT/prof_run.txt | tail -4
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats $(find $OUT/prof -name "*.db" | head -1) | tee $OUT/direct_kernel_stats.txt
rm -rf $OUT/prof
