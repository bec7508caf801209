#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02i
mkdir -p $OUT
cd $R
for lib in femcy_amd/libfemcy_hip.so build/libfemcy_unroll1.so build/libfemcy_unroll3.so build/libfemcy_unroll4.so; do
  for wl in c3d4 c3d10; do
    echo -n "$lib $wl: "
    FEMCY_HIP_LIB=$R/$lib timeout 300 python bench.py --workload $wl --no-cpu-baseline --prewarm 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pcg_us_per_iter %.1f  spmv_us %.1f  value %.0f' % (d['pcg_us_per_iter'], d['roofline']['avg_launch_us'], d['value']))"
  done
done 2>&1 | tee $OUT/unroll.txt
