#!/bin/bash
# round 3, GPU call 4: full GPU suite, bench with the new default persistent variant, PMC traffic of rows2 / rows3
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03d
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=10 > $OUT/pytest_gpu.log 2>&1
tail -16 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
cat $OUT/bench_c3d4.json; tail -3 $OUT/bench_c3d4.err
cd /tmp
for m in 6 7; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_${m}_$ctr -o pmc -- python $R/tools/asm_probe.py c3d10 $m 5 > $OUT/pmc_asm_${m}_$ctr.log 2>&1
    db=$(find $OUT/pmc_${m}_$ctr -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc $db $ctr 2>&1 | grep -E "k_assemble|kernel" > $OUT/pmc_asm_mode${m}_$ctr.txt; fi
    rm -rf $OUT/pmc_${m}_$ctr
  done
done
cat $OUT/pmc_asm_mode*.txt
