#!/bin/bash
# round 6, call 1: CPE8 assembly -- old generic rows kernel (mode 2) vs the pair-list kernel (mode 9), HIP-event times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for mode in 2 9 0 5; do python tools/asm_probe.py cpe8 $mode 30; done 2>&1 | tee gpurun_out/r06_asm_cpe8_probe.txt
