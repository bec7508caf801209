#!/bin/bash
# round 4, GPU call 2: full suite; bench.py as 2 and 4 PROCESSES on one GPU over the shared-memory transport (every host
# step of a multi-GPU run + the cross-process mailbox path); storage order x row order repeated; headline + force-comm
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04b
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/ -x -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2>&1
tail -12 $OUT/pytest_gpu.log
for n in 2 4; do
  FEMCY_BENCH_TRANSPORT=shm FEMCY_BENCH_ALL_ON_GPU0=1 FEMCY_BENCH_DIST_BACKEND=gloo FEMCY_BENCH_DEVICE=cpu GPU_MAX_HW_QUEUES=16 \
    timeout 600 python bench.py --gpus $n --steps 3 --warmup 1 --iters 200 --prewarm 0 --no-cpu-baseline --comm-timeout 120 \
    > $OUT/bench_shm_n$n.json 2> $OUT/bench_shm_n$n.err
  echo "rc $?"; tail -c 3000 $OUT/bench_shm_n$n.json; grep -v "amdgpu.ids\|^\[Gloo\]" $OUT/bench_shm_n$n.err | tail -15
done
for rep in 1 2; do
  timeout 400 python tools/r04_ab.py order c3d10 2>&1 | grep -v amdgpu.ids >> $OUT/ab_order_c3d10.txt
  timeout 500 python tools/r04_ab.py order c3d4_8m 2>&1 | grep -v amdgpu.ids >> $OUT/ab_order_c3d4_8m.txt
done
cat $OUT/ab_order_c3d10.txt $OUT/ab_order_c3d4_8m.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm.json 2> $OUT/bench_forcecomm.err
echo "rc $?"; python -c "
import json,sys
d=json.load(open('$OUT/bench_forcecomm.json'))
print(d['value'], d['config']['persistent_pcg_across_ranks'], d['config']['interface_exchange'])"
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
echo "rc $?"; cat $OUT/bench_c3d4.json | head -c 6000
ls -la $OUT
