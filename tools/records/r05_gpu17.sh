#!/bin/bash
# round 5, call 17: the tile write-out as tuning option 116 -- its parity test, the assembly suites, the timing again
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05q
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pins.py tests/test_gpu_fullsize.py -x -q -m gpu -k "assemble or Ke or c3d10 or C3D10" > $OUT/pytest_asm.log 2>&1; tail -3 $OUT/pytest_asm.log
for cfg in "" "4,19" "4,24" "2,28"; do
  FEMCY_ROWS4_TILE=$cfg timeout 200 python tools/asm_probe.py c3d10 8 30 2>&1 | grep -v amdgpu.ids | sed "s/^/tile [$cfg]: /" >> $OUT/rows4_tile.txt
done
cat $OUT/rows4_tile.txt
