#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03f
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=6 > $OUT/pytest_gpu.log 2>&1
tail -12 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-baseline both > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
cat $OUT/bench_c3d4.json; tail -3 $OUT/bench_c3d4.err
