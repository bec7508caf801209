#!/bin/bash
# round 5, closing call: the whole -m gpu suite on the final sources, then the default bench line
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05z
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
