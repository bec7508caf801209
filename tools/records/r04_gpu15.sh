#!/bin/bash
# round 4, direct solve with the 64 x 64 update: tests, timing, the wide bands
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04direct
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_direct.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "direct or gen_beam or twist_plate_C3D4 or nu0d4999 or ellip_dense or twist_C3D10" > $OUT/pytest3.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest3.log
grep -v "Warn\|^$\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $OUT/pytest3.log | tail -6
timeout 600 python tools/direct_bench.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $OUT/direct_bench.txt
timeout 600 python tools/direct_limit.py 12 20 30 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $OUT/direct_limit.txt
