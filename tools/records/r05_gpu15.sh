#!/bin/bash
# round 5, call 15: k_assemble_rows4 with the write-out from a wave's own LDS tile (FEMCY_ROWS4_TILE=GP,LCUT) against the
# shipped kernel: time (HIP events around the launch, tools/asm_probe.py), then parity under the two main settings
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05o
mkdir -p $OUT
cd $R
for rep in 1 2; do
for cfg in "" "2,28" "4,28" "4,19" "2,19" "4,65"; do
  FEMCY_ROWS4_TILE=$cfg timeout 200 python tools/asm_probe.py c3d10 8 30 2>&1 | grep -v amdgpu.ids | sed "s/^/tile [$cfg]: /" >> $OUT/rows4_tile.txt
done
done
cat $OUT/rows4_tile.txt
for cfg in "2,28" "4,28"; do
  FEMCY_ROWS4_TILE=$cfg timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pins.py tests/test_gpu_fullsize.py -x -q -m gpu -k "assemble or Ke or c3d10 or C3D10" > $OUT/pytest_tile_${cfg/,/_}.log 2>&1; tail -3 $OUT/pytest_tile_${cfg/,/_}.log
done
