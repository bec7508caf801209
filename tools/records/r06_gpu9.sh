#!/bin/bash
# round 6, call 9: k_assemble_rows4 launch order (0 longest first / round-robin, 1 Morton / XCD-contiguous) on the C3D10 plates
# k = 6 (124 k elements: records in the Infinity Cache), 8, 12 (995 k: 0.99 GB of records)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for k in 6 12; do for o in 0 2 3; do FEMCY_PROBE_K=$k FEMCY_PROBE_ROWS4_ORDER=$o python tools/asm_probe.py c3d10 8 20 2>&1 | grep "mode 8\|elements"; done; done | tee gpurun_out/r06_rows4_order.txt
