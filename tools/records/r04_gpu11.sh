#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04i
mkdir -p $OUT
cd $R
for order in 0 1 0 1; do FEMCY_PROBE_NODE_ORDER=$order timeout 200 python tools/asm_probe.py c3d10 8 20 2>&1 | grep -v amdgpu.ids | sed "s/^/node_order $order: /" >> $OUT/rows4_xcd_ranges.txt; done
cat $OUT/rows4_xcd_ranges.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pins.py tests/test_gpu_fullsize.py -x -q -m gpu -k "assemble or Ke or c3d10 or C3D10" 2>&1 | tail -4
