#!/bin/bash
# round 5, call 2: the NaN of call 1 (test_persistent_pcg_four_slices_per_wave) alone in a fresh process; the whole -m gpu
# suite under FEMCY_DEBUG_POISON=1 (every device allocation pre-filled with NaN bytes: uninitialised reads fail
# deterministically); C3D10 workload line with the persistent PCG now the default there; direct branch towards 1e5 DOF
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05b
mkdir -p $OUT
cd $R
timeout 600 python -m pytest "tests/test_gpu_pcg_persist.py::test_persistent_pcg_four_slices_per_wave" -q -m gpu -s > $OUT/pytest_four_alone.log 2>&1; tail -4 $OUT/pytest_four_alone.log
FEMCY_DEBUG_POISON=1 timeout 600 python -m pytest "tests/test_gpu_pcg_persist.py::test_persistent_pcg_four_slices_per_wave" -q -m gpu -s > $OUT/pytest_four_poison.log 2>&1; tail -4 $OUT/pytest_four_poison.log
FEMCY_DEBUG_POISON=1 timeout 1500 python -m pytest tests/ -q -m gpu --deselect tests/test_gpu_cg_branch.py --durations=8 > $OUT/pytest_poison.log 2>&1; tail -30 $OUT/pytest_poison.log
timeout 300 python bench.py --workload c3d10 --steps 10 --no-cpu-baseline > $OUT/bench_c3d10.json 2> $OUT/bench_c3d10.err; tail -2 $OUT/bench_c3d10.err; python -c "
import json;d=json.load(open('$OUT/bench_c3d10.json'));print('c3d10', d['value'], d['pcg_us_per_iter'], d['roofline'])"
timeout 300 python tools/direct_limit.py 12 20 30 2>&1 | grep -v amdgpu.ids > $OUT/direct_limit.txt; cat $OUT/direct_limit.txt
ls -la $OUT
