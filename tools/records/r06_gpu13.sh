#!/bin/bash
# round 6, call 13: bisect the reproduced NaN (first four-slices test of a process that ran the multi-rank file before it)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=tests/test_gpu_pcg_persist.py::test_persistent_pcg_four_slices_per_wave
run() { echo "=== $*"; timeout 300 python -m pytest -q -x -m gpu -p no:cacheprovider "$@" 2>&1 | grep -E "passed|failed|^E  |test_gpu_pcg_persist.py:[0-9]+:" | head -8; }
run tests/test_gpu_multirank.py $T
run $T
run "tests/test_gpu_multirank.py::test_persistent_pcg_across_ranks_times_out_together" $T
run "tests/test_gpu_multirank.py::test_persistent_pcg_across_ranks_on_one_gpu" $T
run "tests/test_gpu_multirank.py::test_partitioned_solve_equals_single_context" $T
run "tests/test_gpu_multirank.py::test_neighbour_exchange_equals_allreduce" $T
run "tests/test_gpu_multirank.py::test_partitioned_deck_solve_equals_single_context" $T
run "tests/test_gpu_multirank.py::test_main_as_one_rank_rccl_job" "tests/test_gpu_multirank.py::test_rendezvous_failures_are_reported" $T
