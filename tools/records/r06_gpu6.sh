#!/bin/bash
# round 6, call 6: the streaming form of k_assemble_pairs (descriptor / codes / first records of the next chunk prefetched):
# parity, then FEMCY_TUNE_PAIRS sweep with chunks per wave (bits 6-9)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_pins.py tests/test_gpu_parity.py -x -q -m gpu -k "pair_list or assemble_K or 2d_element_Ke" 2>&1 | tail -3
python tools/asm_probe.py cpe8 2 10 2>&1 | grep "mode 2"
for k in 0 33 35 97 99 161 163 225 227 291 483 235 171 179 35 99 163 227; do FEMCY_PROBE_PAIRS=$k python tools/asm_probe.py cpe8 9 100 2>&1 | grep "mode 9"; done | tee gpurun_out/r06_asm_cpe8_knobs3.txt
