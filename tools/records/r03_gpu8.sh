#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03h
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pins.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/pytest.log 2>&1
grep -E "passed|failed" $OUT/pytest.log | tail -2
(timeout 300 python tools/microbench.py 12; timeout 300 python tools/microbench.py 6 1) 2>&1 | grep -E "assemble_K|geom kernel|internal_force|spmv wps=0" > $OUT/microbench.txt
cat $OUT/microbench.txt
