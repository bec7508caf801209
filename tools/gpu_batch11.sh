#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02k
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "assemble_K or small" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c3d10 or properties" 2>&1 | tail -3
timeout 120 python tools/asm_probe.py c3d10 6 2>&1 | grep "mode"
timeout 120 python tools/asm_probe.py c3d4 6 2>&1 | grep "mode"
FEMCY_BENCH_RENUM=1 timeout 120 python tools/asm_probe.py c3d10 6 2>&1 | grep "mode"
cd /tmp
for p in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $p -d $OUT/pmc_$p -o pmc -- python $R/tools/asm_probe.py c3d10 6 5 > $OUT/pmc_$p.log 2>&1
  python $R/tools/rocprof_summary.py pmc_all $(find $OUT/pmc_$p -name "*.db" | head -1) k_assemble_rows2
  rm -rf $OUT/pmc_$p
done
