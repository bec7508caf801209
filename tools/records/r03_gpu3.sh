#!/bin/bash
# round 3, GPU call 3: ceilings (finer sizes, more loads in flight), nt / l2-rows variants, rows3 assembly
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03c
mkdir -p $OUT
cd $R
FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_allvar.so ITERS=500 timeout 900 python tools/persist_variants.py c3d4 2>&1 | grep -v "amdgpu.ids" > $OUT/persist_variants_c3d4.txt
cat $OUT/persist_variants_c3d4.txt
timeout 900 python -m pytest tests/test_gpu_pcg_persist.py -x -q -m gpu > $OUT/pytest_persist.log 2>&1
tail -5 $OUT/pytest_persist.log
FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_allvar.so timeout 600 python -m pytest tests/test_gpu_pcg_persist.py -q -m gpu -k "variants_agree or exchange_timeout or probes or barrier_timeout" > $OUT/pytest_allvar.log 2>&1
tail -5 $OUT/pytest_allvar.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pins.py -x -q -m gpu -k "assemble_K or exact or one_element" > $OUT/pytest_asm.log 2>&1
tail -5 $OUT/pytest_asm.log
for m in 6 7; do timeout 300 python tools/asm_probe.py c3d10 $m 20 2>&1 | grep "mode"; done > $OUT/asm_probe.txt
for m in 6 7; do timeout 300 python tools/asm_probe.py c3d4 $m 20 2>&1 | grep "mode"; done >> $OUT/asm_probe.txt
cat $OUT/asm_probe.txt
