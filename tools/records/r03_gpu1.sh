#!/bin/bash
# round 3, GPU call 1: new tests, persistent-PCG variants + ceilings, the new bench line
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_pcg_persist.py tests/test_gpu_bench_contract.py -x -q -m gpu --durations=8 > $OUT/pytest_persist.log 2>&1
tail -15 $OUT/pytest_persist.log
FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_allvar.so timeout 600 python -m pytest tests/test_gpu_pcg_persist.py -x -q -m gpu -k "variants_agree or exchange_timeout or probes" > $OUT/pytest_allvar.log 2>&1
tail -8 $OUT/pytest_allvar.log
FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_allvar.so ITERS=500 timeout 600 python tools/persist_variants.py c3d4 2>&1 | grep -v "amdgpu.ids" > $OUT/persist_variants_c3d4.txt
cat $OUT/persist_variants_c3d4.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
cat $OUT/bench_c3d4.json; tail -5 $OUT/bench_c3d4.err
