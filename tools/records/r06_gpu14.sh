#!/bin/bash
# round 6, call 14: the pair-list assembly instantiated for the 3-D families: parity, then time against the shipped kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pins.py -q -m gpu -k "assemble_K or Ke_equals or unreferenced" 2>&1 | tail -3
(for wl in c3d4 c3d10; do for mode in 3 9; do python tools/asm_probe.py $wl $mode 30 2>&1 | grep "mode $mode"; done; for k in 161 163 99 97 227; do FEMCY_PROBE_PAIRS=$k python tools/asm_probe.py $wl 9 30 2>&1 | grep "mode 9"; done; done) | tee gpurun_out/r06_pairs_3d.txt
