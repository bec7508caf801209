#!/bin/bash
# round 4: the new pins and property tests on the GPU
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04pins
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pins.py tests/test_gpu_direct.py tests/test_gpu_fullsize.py -m gpu -q -s -p no:cacheprovider -k "gradient or new_pattern or cuthill" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "Warn\|^$\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $OUT/pytest.log | tail -25
