#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02d
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "assemble_K" > $OUT/pytest_asm.log 2>&1
tail -3 $OUT/pytest_asm.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c3d10 or properties" > $OUT/pytest_full.log 2>&1
tail -3 $OUT/pytest_full.log
timeout 600 python -m pytest tests/test_gpu_pins.py -x -q -m gpu > $OUT/pytest_pins.log 2>&1
tail -15 $OUT/pytest_pins.log
rm -f $OUT/probe.txt
timeout 120 python tools/asm_probe.py c3d10 2>&1 | grep "mode" >> $OUT/probe.txt
timeout 120 python tools/asm_probe.py c3d4 6 2>&1 | grep "mode" >> $OUT/probe.txt
cat $OUT/probe.txt
