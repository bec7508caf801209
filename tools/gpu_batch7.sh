#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02g
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_parity.log 2>&1
tail -4 $OUT/pytest_parity.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_pins.py -x -q -m gpu > $OUT/pytest_full.log 2>&1
tail -4 $OUT/pytest_full.log
for m in 5 4 0 6; do timeout 120 python tools/asm_probe.py c3d4 $m 2>&1 | grep "mode"; done > $OUT/probe.txt
timeout 120 python tools/asm_probe.py c3d10 6 2>&1 | grep "mode" >> $OUT/probe.txt
cat $OUT/probe.txt
timeout 300 python bench.py --workload c3d10 --no-cpu-baseline --prewarm 1 > $OUT/bench_c3d10.json 2>/dev/null; cat $OUT/bench_c3d10.json
