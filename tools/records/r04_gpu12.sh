#!/bin/bash
# round 4, direct solve: its tests, the decks through the driver, timing against the tight PCG
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04direct
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_e2e.py -m gpu -q -s -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "Warn\|^$" $OUT/pytest.log | tail -45
timeout 600 python tools/direct_bench.py > $OUT/direct_bench.txt 2>&1
cat $OUT/direct_bench.txt | grep -v Warn
