#!/bin/bash
# round 6, call 3: FEMCY_TUNE_PAIRS sweep on the CPE8 beam (bit 0 XCD-contiguous, bits 1-2 rows per wave, bits 3-4 depth - 2)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for k in 0 1 2 3 8 9 10 11 16 17 18 19; do FEMCY_PROBE_PAIRS=$k python tools/asm_probe.py cpe8 9 30 2>&1 | grep "mode 9"; done | tee gpurun_out/r06_asm_cpe8_knobs.txt
