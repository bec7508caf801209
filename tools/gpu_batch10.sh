#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02j
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pcg or spmv" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for lib in build/libfemcy_head.so femcy_amd/libfemcy_hip.so; do
  for wl in c3d4 c3d10; do
    echo -n "$lib $wl: "
    FEMCY_HIP_LIB=$R/$lib timeout 300 python bench.py --workload $wl --no-cpu-baseline --prewarm 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pcg_us_per_iter %.1f  spmv_us %.1f  value %.0f' % (d['pcg_us_per_iter'], d['roofline']['avg_launch_us'], d['value']))"
  done
done; done 2>&1 | tee $OUT/xcd_vec.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $R/bench.py --steps 3 --no-cpu-baseline --prewarm 1 > $OUT/kt.log 2>&1
python $R/tools/rocprof_summary.py stats $(find $OUT/kt -name "*.db" | head -1) | head -6
rm -rf $OUT/kt
