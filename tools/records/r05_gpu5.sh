#!/bin/bash
# round 5, call 5: the CG-branch tests against the completed fixture, the new bench contract tests, the re-timed
# four-slices test
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05e
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_cg_branch.py -q -m gpu -s > $OUT/pytest_cg_branch.log 2>&1; grep -v "^  File\|^Thread" $OUT/pytest_cg_branch.log | tail -40
timeout 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_pcg_persist.py -q -m gpu -s -k "cpe8 or 2d_configuration or four_slices" > $OUT/pytest_contract.log 2>&1; tail -12 $OUT/pytest_contract.log
ls -la $OUT
