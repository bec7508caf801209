#!/bin/bash
# experimental builds of ONE (or two) translation unit(s): tools/build_variant.sh <name> <file under femcy_amd/csrc> "<extra flags>" [second file]
# -> femcy_amd/libfemcy_hip_<name>.so (select with FEMCY_HIP_LIB).  The other objects are compiled once into build/obj
# (delete that directory after touching their sources; femcy_amd/csrc/build.sh stays the build of the shipped library).
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; file=$2; extra=${3:-}; file2=${4:-}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-result -x hip"
mkdir -p $R/build/obj
cd $R/femcy_amd/csrc
objs=""
for f in femcy_api.cpp pattern.cpp comm.cpp kernels_assembly.hip kernels_pcg.hip kernels_pcg_persist.hip kernels_direct.hip; do
  o=$R/build/obj/${f%.*}.o
  if [ "$f" = "$file" ] || [ "$f" = "$file2" ]; then
    o=$R/build/obj/${f%.*}_$name.o
    $HIPCC $FLAGS $extra -c $f -o $o
  elif [ ! -f $o ] || [ $f -nt $o ]; then
    $HIPCC $FLAGS -c $f -o $o
  fi
  objs="$objs $o"
done
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o $R/femcy_amd/libfemcy_hip_$name.so -Wl,-rpath,/opt/rocm/lib -ldl -lpthread
echo "built femcy_amd/libfemcy_hip_$name.so"
