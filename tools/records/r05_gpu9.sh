#!/bin/bash
# round 5, call 9: the refinement threshold of femcy_direct_solve (1e-12: one refinement solve = a forward + backward
# sweep, 8 us per panel, on almost every system) against 1e-10 (-DFEMCY_DIRECT_REFINE_ABOVE=1e-10): time per solve, and the
# 49 decks end to end on the looser build
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05i
mkdir -p $OUT
cd $R
(REPS=5 timeout 300 python tools/direct_bench.py; FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_refine10.so REPS=5 timeout 300 python tools/direct_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/direct_bench_refine.txt; cat $OUT/direct_bench_refine.txt
(FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_refine10.so timeout 300 python tools/direct_limit.py 12 20 30) 2>&1 | grep -v amdgpu.ids > $OUT/direct_limit_refine10.txt; cat $OUT/direct_limit_refine10.txt
FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_refine10.so timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_direct.py -q -m gpu -s > $OUT/pytest_refine10.log 2>&1; grep "rel L2" $OUT/pytest_refine10.log | awk '{print $1, $5}' | sort -k2 -g | tail -5; tail -3 $OUT/pytest_refine10.log
