#!/bin/bash
# round 5, call 13: the reference's CG leg on its own decks (48 decks, cg_branch_from = 0) against the oracle's as-written CG
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05m
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_cg_branch.py -q -m gpu -s -k "cg_leg" > $OUT/pytest_cg_leg.log 2>&1; grep "cg leg" $OUT/pytest_cg_leg.log | cut -c1-230; tail -3 $OUT/pytest_cg_leg.log
