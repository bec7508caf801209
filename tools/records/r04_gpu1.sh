#!/bin/bash
# round 4, GPU call 1: the full GPU suite on the new defaults (storage-order three-launch PCG), the cross-process
# tests, A/B of the persistent PCG's in-band variant (two library builds) and of storage order x row order
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04a
mkdir -p $OUT
cd $R
rocminfo 2>/dev/null | grep -c "gfx950" > $OUT/ngpu.txt
timeout 900 python -m pytest tests/ -x -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2>&1
tail -25 $OUT/pytest_gpu.log
timeout 600 python -m pytest tests/test_gpu_xproc.py -q -m gpu > $OUT/pytest_xproc.log 2>&1
tail -30 $OUT/pytest_xproc.log
timeout 300 python tools/r04_ab.py persist 2>&1 | grep -v amdgpu.ids > $OUT/ab_persist_default.txt
FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_inbnp.so timeout 300 python tools/r04_ab.py persist 2>&1 | grep -v amdgpu.ids > $OUT/ab_persist_inbnp.txt
cat $OUT/ab_persist_default.txt $OUT/ab_persist_inbnp.txt
timeout 400 python tools/r04_ab.py order c3d10 2>&1 | grep -v amdgpu.ids > $OUT/ab_order_c3d10.txt
cat $OUT/ab_order_c3d10.txt
timeout 500 python tools/r04_ab.py order c3d4_8m 2>&1 | grep -v amdgpu.ids > $OUT/ab_order_c3d4_8m.txt
cat $OUT/ab_order_c3d4_8m.txt
ls -la $OUT
