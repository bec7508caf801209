#!/bin/bash
# round 6, call 5: where the pair-list assembly's time goes -- work-skipping builds (-DFEMCY_PAIRS_PROBE bits: 1 no global
# stores, 2 no LDS atomics, 4 no record loads, 8 no tile zeroing / final LDS reads, 16 no cross-lane reads), knobs 33 and 35
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for k in 33 35; do
  FEMCY_PROBE_PAIRS=$k python tools/asm_probe.py cpe8 9 30 2>&1 | grep "mode 9"
  for b in 1 2 4 8 16 5 7 31; do FEMCY_HIP_LIB=$PWD/femcy_amd/libfemcy_hip_pp$b.so FEMCY_PROBE_PAIRS=$k python tools/asm_probe.py cpe8 9 30 2>&1 | grep "mode 9"; done
done | tee gpurun_out/r06_pairs_probe.txt
