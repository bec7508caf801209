#!/bin/bash
# Build libfemcy_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${FEMCY_OUT:-../libfemcy_hip.so}
SRC="femcy_api.cpp pattern.cpp comm.cpp kernels_assembly.hip kernels_pcg.hip kernels_pcg_persist.hip kernels_direct.hip"
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics \
    -Wall -Wno-unused-result -x hip $SRC -o $OUT \
    -Wl,-rpath,/opt/rocm/lib -ldl -lpthread ${FEMCY_EXTRA_FLAGS:-}
echo "built $(realpath $OUT)"
