#!/bin/bash
# round 4, GPU call 3: full suite; bench.py as 2 / 4 processes on one GPU (shared-memory transport) with per-rank meshes
# that fit a share of the CUs; launch-shape knobs of the storage-order product; quick force-comm check of the fused
# pair reduction
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04c
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/ -x -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2>&1
tail -12 $OUT/pytest_gpu.log
for n in 2 4; do
  FEMCY_BENCH_TRANSPORT=shm FEMCY_BENCH_ALL_ON_GPU0=1 FEMCY_BENCH_DIST_BACKEND=gloo FEMCY_BENCH_DEVICE=cpu GPU_MAX_HW_QUEUES=16 \
    FEMCY_BENCH_STRONG_CELLS=96,12,144 \
    timeout 600 python bench.py --gpus $n --cells 48,12,144 --steps 3 --warmup 1 --iters 200 --prewarm 0 --no-cpu-baseline --comm-timeout 120 \
    > $OUT/bench_shm_n$n.json 2> $OUT/bench_shm_n$n.err
  echo "rc $?"; python -c "
import json
d=json.load(open('$OUT/bench_shm_n$n.json'))
print('weak-style', d['value'], d['pcg_us_per_iter'], json.dumps(d['config']['persistent_pcg_across_ranks']))
print('strong', json.dumps(d.get('strong_scaling'))[:1500])"
done
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm.json 2> $OUT/bench_forcecomm.err
FEMCY_BENCH_PERSIST_MULTI=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm_loop.json 2> $OUT/bench_forcecomm_loop.err
python -c "
import json
for f in ('bench_forcecomm','bench_forcecomm_loop'):
    d=json.load(open('$OUT/'+f+'.json')); print(f, d['value'], d['pcg_us_per_iter'], json.dumps(d['config']['persistent_pcg_across_ranks']))"
timeout 400 python tools/r04_ab.py knobs c3d10 2>&1 | grep -v amdgpu.ids > $OUT/knobs_c3d10.txt; cat $OUT/knobs_c3d10.txt
timeout 400 python tools/r04_ab.py knobs c3d4_8m 2>&1 | grep -v amdgpu.ids > $OUT/knobs_c3d4_8m.txt; cat $OUT/knobs_c3d4_8m.txt
ls -la $OUT
