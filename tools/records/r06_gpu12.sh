#!/bin/bash
# round 6, call 12: the NaN hunt (tools/r06_nan_hunt.py): 10 x (multi-rank file -> four-slices test) + 50 x context churn in one
# process, once plain and once under FEMCY_DEBUG_POISON=1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(timeout 2400 python tools/r06_nan_hunt.py 10 50 2>&1 | grep "nan hunt\|passed\|failed\|Error" ) > gpurun_out/r06_nan_hunt_plain.txt
(FEMCY_DEBUG_POISON=1 timeout 2400 python tools/r06_nan_hunt.py 6 30 2>&1 | grep "nan hunt\|passed\|failed\|Error") > gpurun_out/r06_nan_hunt_poison.txt
tail -4 gpurun_out/r06_nan_hunt_plain.txt; tail -4 gpurun_out/r06_nan_hunt_poison.txt
