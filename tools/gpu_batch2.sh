#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "assemble_K or residual_and_K or internal_force or postprocessing" > $OUT/pytest_asm.log 2>&1
tail -5 $OUT/pytest_asm.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c3d10 or properties" > $OUT/pytest_full.log 2>&1
tail -5 $OUT/pytest_full.log
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_tangent.py tests/test_gpu_neohooke2d.py -x -q -m gpu > $OUT/pytest_e2e.log 2>&1
tail -5 $OUT/pytest_e2e.log
(timeout 300 python tools/microbench.py 6 1; timeout 300 python tools/microbench.py 12) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | grep "assemble\|geom\|internal" > $OUT/microbench.txt
cat $OUT/microbench.txt
