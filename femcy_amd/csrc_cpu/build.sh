#!/bin/bash
# Build libfemcy_cpu.so (the host backend of include/femcy.h) in-tree.  x86-64-v3 (AVX2 + FMA), not -march=native: the
# library is built in the build container and travels to other boxes.
set -euo pipefail
cd "$(dirname "$0")"
CXX=${CXX:-g++}
OUT=${FEMCY_CPU_OUT:-../libfemcy_cpu.so}
$CXX -O3 -march=x86-64-v3 -std=c++17 -fopenmp -fPIC -shared -Wall -Wno-unknown-pragmas -Wno-unused-function \
    femcy_cpu.cpp -o $OUT
echo "built $(realpath $OUT)"
