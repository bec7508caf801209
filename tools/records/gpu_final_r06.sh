#!/bin/bash
# round-6 record run (one MI355X): smoke, the three workload lines, the forced-communicator lines, 2- and 8-process rehearsals
# of `bench.py --gpus N` on one GPU over the shared-memory transport, kernel traces, PMC passes (FETCH / WRITE ->
# profiles/spmv_traffic.json, stamped with the machine code of the PCG kernels and the layout of each run), test suites
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06final
HEAD_SHA=${1:-unknown}
mkdir -p $OUT
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
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
spec() { echo "$1:$(find $OUT/fetch_$1 -name '*.db' | head -1):$(find $OUT/write_$1 -name '*.db' | head -1):$OUT/fetch_$1.log"; }
python tools/make_traffic_json.py $HEAD_SHA $(spec c3d4) $(spec c3d10) $(spec cpe8) > $OUT/traffic.log 2>&1
cp profiles/spmv_traffic.json $OUT/spmv_traffic.json
rm -rf $OUT/kt_c3d4 $OUT/kt_c3d10 $OUT/kt_cpe8 $OUT/fetch_c3d4 $OUT/fetch_c3d10 $OUT/fetch_cpe8 $OUT/write_c3d4 $OUT/write_c3d10 $OUT/write_cpe8
timeout 300 python bench.py --workload c3d10 --steps 10 --no-cpu-baseline > $OUT/bench_c3d10.json 2> $OUT/bench_c3d10.err
timeout 400 python bench.py --workload cpe8 --steps 10 --no-cpu-baseline > $OUT/bench_cpe8.json 2> $OUT/bench_cpe8.err
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
cat $OUT/bench_c3d4.json | cut -c1-400
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm_persistent.json 2> $OUT/bench_forcecomm_persistent.err
FEMCY_BENCH_PERSIST_MULTI=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm_rccl.json 2> $OUT/bench_forcecomm_rccl.err
FEMCY_BENCH_PERSIST=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off > $OUT/bench_c3d4_three_kernel.json 2> $OUT/bench_c3d4_three_kernel.err
for spec in "2 48,12,144" "8 48,12,288"; do
  set -- $spec
  FEMCY_BENCH_TRANSPORT=shm FEMCY_BENCH_ALL_ON_GPU0=1 FEMCY_BENCH_DIST_BACKEND=gloo FEMCY_BENCH_DEVICE=cpu GPU_MAX_HW_QUEUES=16 \
    FEMCY_BENCH_STRONG_CELLS=48,12,144 \
    timeout 900 python bench.py --gpus $1 --cells $2 --steps 3 --warmup 1 --iters 200 --prewarm 0 --no-cpu-baseline --comm-timeout 240 \
    > $OUT/bench_shm_n$1.json 2> $OUT/bench_shm_n$1.err
  tail -c 600 $OUT/bench_shm_n$1.json; tail -3 $OUT/bench_shm_n$1.err
done
FEMCY_DEBUG_POISON=1 timeout 1800 python -m pytest tests/ -q -m gpu -p no:faulthandler > $OUT/pytest_gpu_poison.log 2>&1; tail -3 $OUT/pytest_gpu_poison.log
timeout 1800 python -m pytest tests/ -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
ls -la $OUT
