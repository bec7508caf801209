#!/bin/bash
# round-5 record run, second half (after the last source changes: public SpMV back on the node-order kernel, poison fill on
# its own stream): PMC passes -> profiles/spmv_traffic.json, the three workload lines with live traffic, the poisoned suite
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05final
HEAD_SHA=${1:-unknown}
mkdir -p $OUT
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
timeout 300 python bench.py --workload c3d10 --steps 10 --no-cpu-baseline > $OUT/bench_c3d10.json 2> $OUT/bench_c3d10.err
timeout 400 python bench.py --workload cpe8 --steps 10 --no-cpu-baseline > $OUT/bench_cpe8.json 2> $OUT/bench_cpe8.err
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
cat $OUT/bench_c3d4.json | cut -c1-600
FEMCY_DEBUG_POISON=1 timeout 1500 python -m pytest tests/ -q -m gpu -p no:faulthandler > $OUT/pytest_gpu_poison.log 2>&1; tail -3 $OUT/pytest_gpu_poison.log
timeout 1500 python -m pytest tests/ -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
ls -la $OUT
