#!/bin/bash
# round 3, GPU call 2: persistent-PCG variants + ceilings (all-variants build), persist tests with both libraries
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03b
mkdir -p $OUT
cd $R
FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_allvar.so ITERS=500 timeout 600 python tools/persist_variants.py c3d4 2>&1 | grep -v "amdgpu.ids" > $OUT/persist_variants_c3d4.txt
cat $OUT/persist_variants_c3d4.txt
timeout 900 python -m pytest tests/test_gpu_pcg_persist.py -x -q -m gpu > $OUT/pytest_persist.log 2>&1
tail -8 $OUT/pytest_persist.log
FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_allvar.so timeout 600 python -m pytest tests/test_gpu_pcg_persist.py -q -m gpu -k "variants_agree or exchange_timeout or probes or barrier_timeout" > $OUT/pytest_allvar.log 2>&1
tail -8 $OUT/pytest_allvar.log
