#!/bin/bash
# round-5 record run: full GPU suite (plain and under FEMCY_DEBUG_POISON=1), smoke, bench lines (headline with hbm_bound x 3
# + cpu_baseline; three-kernel; C3D10 persistent / three launches; CPE8; 1-rank RCCL communicator both ways; 2 processes on
# one GPU over the shared-memory transport), kernel traces, PMC passes (FETCH / WRITE -> profiles/spmv_traffic.json)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05final
HEAD_SHA=${1:-unknown}
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1
tail -16 $OUT/pytest_gpu.log
FEMCY_DEBUG_POISON=1 timeout 1500 python -m pytest tests/ -q -m gpu -p no:faulthandler > $OUT/pytest_gpu_poison.log 2>&1; tail -3 $OUT/pytest_gpu_poison.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
FEMCY_BENCH_PERSIST=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off > $OUT/bench_c3d4_three_kernel.json 2> $OUT/bench_c3d4_three_kernel.err
FEMCY_BENCH_PERSIST=0 timeout 300 python bench.py --workload c3d10 --steps 10 --no-cpu-baseline > $OUT/bench_c3d10_three_kernel.json 2> $OUT/bench_c3d10_three_kernel.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm_persistent.json 2> $OUT/bench_forcecomm_persistent.err
FEMCY_BENCH_PERSIST_MULTI=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm_rccl.json 2> $OUT/bench_forcecomm_rccl.err
for n in 2; do
  FEMCY_BENCH_TRANSPORT=shm FEMCY_BENCH_ALL_ON_GPU0=1 FEMCY_BENCH_DIST_BACKEND=gloo FEMCY_BENCH_DEVICE=cpu GPU_MAX_HW_QUEUES=16 \
    FEMCY_BENCH_STRONG_CELLS=96,12,144 \
    timeout 600 python bench.py --gpus $n --cells 48,12,144 --steps 3 --warmup 1 --iters 200 --prewarm 0 --no-cpu-baseline --comm-timeout 120 \
    > $OUT/bench_shm_n$n.json 2> $OUT/bench_shm_n$n.err
done
cd /tmp
for wl in c3d4 c3d10 cpe8; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$wl -o kt -- python $R/bench.py --workload $wl --steps 3 --no-cpu-baseline --hbm-bound off --prewarm 1 > $OUT/kt_$wl.log 2>&1
  python $R/tools/rocprof_summary.py stats $(find $OUT/kt_$wl -name "*.db" | head -1) > $OUT/kernel_stats_$wl.txt 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$wl -o pmc -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --hbm-bound off > $OUT/fetch_$wl.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$wl -o pmc -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --hbm-bound off > $OUT/write_$wl.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find $OUT/fetch_$wl -name "*.db" | head -1) FETCH_SIZE > $OUT/pmc_fetch_$wl.txt 2>&1
  python $R/tools/rocprof_summary.py pmc $(find $OUT/write_$wl -name "*.db" | head -1) WRITE_SIZE > $OUT/pmc_write_$wl.txt 2>&1
done
cd $R
python tools/make_traffic_json.py $HEAD_SHA c3d4:$(find $OUT/fetch_c3d4 -name "*.db" | head -1):$(find $OUT/write_c3d4 -name "*.db" | head -1) c3d10:$(find $OUT/fetch_c3d10 -name "*.db" | head -1):$(find $OUT/write_c3d10 -name "*.db" | head -1) cpe8:$(find $OUT/fetch_cpe8 -name "*.db" | head -1):$(find $OUT/write_cpe8 -name "*.db" | head -1) > $OUT/traffic.log 2>&1
cp profiles/spmv_traffic.json $OUT/spmv_traffic.json
rm -rf $OUT/kt_c3d4 $OUT/kt_c3d10 $OUT/kt_cpe8 $OUT/fetch_c3d4 $OUT/fetch_c3d10 $OUT/fetch_cpe8 $OUT/write_c3d4 $OUT/write_c3d10 $OUT/write_cpe8
head -8 $OUT/kernel_stats_c3d4.txt; head -8 $OUT/kernel_stats_c3d10.txt; head -8 $OUT/kernel_stats_cpe8.txt
timeout 300 python bench.py --workload c3d10 --steps 10 --no-cpu-baseline > $OUT/bench_c3d10.json 2> $OUT/bench_c3d10.err
timeout 400 python bench.py --workload cpe8 --steps 10 --no-cpu-baseline > $OUT/bench_cpe8.json 2> $OUT/bench_cpe8.err
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
cat $OUT/bench_c3d4.json | cut -c1-1500
ls -la $OUT
