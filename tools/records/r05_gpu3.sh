#!/bin/bash
# round 5, call 3: the poisoned-allocation fault located step by step (kernels serialised); the new persistent shapes for
# 2 x 2 blocks (6 / 8 slices per wave); direct = "auto"; the 2-D bench line on the persistent path
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05c
mkdir -p $OUT
cd $R
for wl in c3d4_small c3d4 c3d10; do
  FEMCY_DEBUG_POISON=1 AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 timeout 300 python tools/r05_poison_probe.py $wl 2>&1 | grep -v "amdgpu.ids\|^  File\|^Thread\|^Extension" | head -40 > $OUT/poison_probe_$wl.txt
  cat $OUT/poison_probe_$wl.txt
done
timeout 300 python tools/r05_poison_probe.py c3d10 2>&1 | grep -v "amdgpu.ids" | head -40 > $OUT/probe_c3d10_nopoison.txt; cat $OUT/probe_c3d10_nopoison.txt
timeout 900 python -m pytest tests/test_gpu_pcg_persist.py -q -m gpu -s -k "2d_six or four_slices or two_dimensional" > $OUT/pytest_persist2d.log 2>&1; grep -v "^  File\|^Thread" $OUT/pytest_persist2d.log | tail -25
timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_e2e.py -q -m gpu -s -k "auto or driver_takes or readme" > $OUT/pytest_auto.log 2>&1; tail -12 $OUT/pytest_auto.log
timeout 400 python bench.py --workload cpe8 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cpe8.json 2> $OUT/bench_cpe8.err; tail -3 $OUT/bench_cpe8.err; python -c "
import json;d=json.load(open('$OUT/bench_cpe8.json'));print('cpe8', d['value'], d['pcg_us_per_iter'], d['assembly_ms'], d['hbm_bound'][0]['pcg_iteration'], d['roofline']['frac'], d['roofline']['through_femcy_spmv_frac'])"
ls -la $OUT
