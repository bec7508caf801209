#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02e
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_multi.log 2>&1
tail -5 $OUT/pytest_multi.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/pytest_full.log 2>&1
tail -5 $OUT/pytest_full.log
timeout 120 python tools/asm_probe.py c3d10 2>&1 | grep "mode" > $OUT/probe.txt
cat $OUT/probe.txt
for ex in allreduce neighbour; do
  timeout 300 python bench.py --force-comm --exchange $ex --no-cpu-baseline --prewarm 1 > $OUT/bench_forcecomm_$ex.json 2> $OUT/bench_forcecomm_$ex.err
  cat $OUT/bench_forcecomm_$ex.json
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $R/bench.py --force-comm --exchange neighbour --no-cpu-baseline --prewarm 0 --steps 2 --warmup 1 > $OUT/kt.log 2>&1
python $R/tools/rocprof_summary.py stats $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats_forcecomm_neighbour.txt 2>&1
rm -rf $OUT/kt
head -30 $OUT/kernel_stats_forcecomm_neighbour.txt
