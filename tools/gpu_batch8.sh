#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02h
mkdir -p $OUT
cd $R
for rn in 0 1; do for sg in 4096 32768; do
  echo "== renum $rn sigma $sg"
  FEMCY_BENCH_RENUM=$rn FEMCY_BENCH_SIGMA=$sg timeout 120 python tools/asm_probe.py c3d10 6 2>&1 | grep "mode"
  FEMCY_BENCH_RENUM=$rn FEMCY_BENCH_SIGMA=$sg timeout 300 python bench.py --workload c3d10 --no-cpu-baseline --prewarm 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench: pcg_us_per_iter %.1f  spmv_us %.1f  assembly_ms %.3f' % (d['pcg_us_per_iter'], d['roofline']['avg_launch_us'], d['assembly_ms']))"
done; done 2>&1 | tee $OUT/ab.txt
for sg in 4096 32768; do echo "== c3d4 sigma $sg"; FEMCY_BENCH_SIGMA=$sg timeout 300 python bench.py --no-cpu-baseline --prewarm 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench: pcg_us_per_iter %.1f  spmv_us %.1f  assembly_ms %.3f value %.0f' % (d['pcg_us_per_iter'], d['roofline']['avg_launch_us'], d['assembly_ms'], d['value']))"; done 2>&1 | tee -a $OUT/ab.txt
