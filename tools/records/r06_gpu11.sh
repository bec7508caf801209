#!/bin/bash
# round 6, call 11: the round-5 tile write-out of k_assemble_rows4 (whole 64 / 128-byte runs per slot instead of 32) re-measured
# OUT OF CACHE (k = 12: 2.8 GB of K), where the read-for-fill of partially written lines is HBM traffic
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for k in 6 12; do for t in "" "2,28" "4,28" "4,40" "2,65" ; do FEMCY_PROBE_K=$k FEMCY_ROWS4_TILE=$t python tools/asm_probe.py c3d10 8 20 2>&1 | grep "mode 8" | sed "s/$/ tile [$t]/"; done; done | tee gpurun_out/r06_rows4_tile_k12.txt
