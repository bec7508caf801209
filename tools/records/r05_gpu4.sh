#!/bin/bash
# round 5, call 4: the three-launch anomaly of the C3D10 section of test_persistent_pcg_four_slices_per_wave (204-235 us per
# iteration against 78 elsewhere; a NaN once) probed; the whole -m gpu suite under FEMCY_DEBUG_POISON=1 (fill synchronised)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05d
mkdir -p $OUT
cd $R
(timeout 300 python tools/r05_three_launch_probe.py zero; timeout 300 python tools/r05_three_launch_probe.py s1) 2>&1 | grep -v amdgpu.ids > $OUT/three_launch_probe.txt; cat $OUT/three_launch_probe.txt
FEMCY_DEBUG_POISON=1 timeout 1800 python -m pytest tests/ -q -m gpu --deselect tests/test_gpu_cg_branch.py -rf --durations=8 -p no:faulthandler > $OUT/pytest_poison.log 2>&1; grep -v "^  File\|^Thread" $OUT/pytest_poison.log | tail -40
ls -la $OUT
