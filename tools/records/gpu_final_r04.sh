#!/bin/bash
# round-4 record run: full GPU suite, smoke, bench lines (headline with hbm_bound + cpu_baseline; three-kernel; C3D10;
# 1-rank RCCL communicator both ways; 2 / 4 PROCESSES on one GPU over the shared-memory transport), A/B records
# (persistent PCG variants incl. the in-band build, storage order x row order, launch-shape knobs), kernel traces, PMC
# passes (bench FETCH / WRITE -> profiles/spmv_traffic.json; SQ / TA / TCC counters of the C3D10 product in node and in
# storage order)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04final
HEAD_SHA=${1:-unknown}
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1
tail -16 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
FEMCY_BENCH_PERSIST=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off > $OUT/bench_c3d4_three_kernel.json 2> $OUT/bench_c3d4_three_kernel.err
timeout 300 python bench.py --workload c3d10 --steps 10 --no-cpu-baseline > $OUT/bench_c3d10.json 2> $OUT/bench_c3d10.err
FEMCY_BENCH_STORAGE_ORDER=0 timeout 300 python bench.py --workload c3d10 --steps 10 --no-cpu-baseline > $OUT/bench_c3d10_node_order_vectors.json 2> $OUT/bench_c3d10_nov.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm_persistent.json 2> $OUT/bench_forcecomm_persistent.err
FEMCY_BENCH_PERSIST_MULTI=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm_rccl.json 2> $OUT/bench_forcecomm_rccl.err
timeout 300 python bench.py --steps 3 --warmup 1 --iters 200 --prewarm 0 --no-cpu-baseline --hbm-bound off --force-comm --force-dist --no-strong > $OUT/bench_forcecomm_forcedist.json 2> $OUT/bench_forcecomm_forcedist.err; echo "force-dist rc $?"
for n in 2 4; do
  FEMCY_BENCH_TRANSPORT=shm FEMCY_BENCH_ALL_ON_GPU0=1 FEMCY_BENCH_DIST_BACKEND=gloo FEMCY_BENCH_DEVICE=cpu GPU_MAX_HW_QUEUES=16 \
    FEMCY_BENCH_STRONG_CELLS=96,12,144 \
    timeout 600 python bench.py --gpus $n --cells 48,12,144 --steps 3 --warmup 1 --iters 200 --prewarm 0 --no-cpu-baseline --comm-timeout 120 \
    > $OUT/bench_shm_n$n.json 2> $OUT/bench_shm_n$n.err
done
# libraries built beside the shipped one (csrc/build.sh with FEMCY_EXTRA_FLAGS / FEMCY_OUT): _inbnp = -DFEMCY_INB_PREFETCH=0 (variant 14
# without the prefetch batch), _noowndiag = -DFEMCY_PERSIST_OWN_DIAG=0 (the diagonal block's d gathered like the others)
(timeout 300 python tools/r04_ab.py persist; FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_inbnp.so timeout 300 python tools/r04_ab.py persist; \
 FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_noowndiag.so timeout 300 python tools/r04_ab.py persist) 2>&1 | grep -v "amdgpu.ids\|^+ " > $OUT/persist_inband.txt
cat $OUT/persist_inband.txt
(timeout 400 python tools/r04_ab.py order c3d10; timeout 400 python tools/r04_ab.py order c3d10) 2>&1 | grep -v amdgpu.ids > $OUT/ab_order_c3d10.txt
(timeout 500 python tools/r04_ab.py order c3d4_8m; timeout 500 python tools/r04_ab.py order c3d4_8m) 2>&1 | grep -v amdgpu.ids > $OUT/ab_order_c3d4_8m.txt
(timeout 400 python tools/r04_ab.py knobs c3d10; timeout 400 python tools/r04_ab.py knobs c3d4_8m) 2>&1 | grep -v amdgpu.ids > $OUT/spmv_knobs.txt
(for wl in c3d10 c3d4 c3d4_8m; do timeout 400 python tools/r04_ab.py fused $wl; done) 2>&1 | grep -v "amdgpu.ids\|^+ " > $OUT/ab_fused.txt
(for wl in c3d4 c3d10 c3d4_8m; do timeout 500 python tools/r04_ab.py footprint $wl; done) 2>&1 | grep -v "amdgpu.ids\|^+ " > $OUT/ab_footprint.txt
(timeout 300 python tools/microbench.py 12; timeout 300 python tools/microbench.py 6 1) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $OUT/microbench.txt
cd /tmp
for wl in c3d4 c3d10; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$wl -o kt -- python $R/bench.py --workload $wl --steps 3 --no-cpu-baseline --hbm-bound off --prewarm 1 > $OUT/kt_$wl.log 2>&1
  python $R/tools/rocprof_summary.py stats $(find $OUT/kt_$wl -name "*.db" | head -1) > $OUT/kernel_stats_$wl.txt 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$wl -o pmc -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --hbm-bound off > $OUT/fetch_$wl.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$wl -o pmc -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --hbm-bound off > $OUT/write_$wl.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find $OUT/fetch_$wl -name "*.db" | head -1) FETCH_SIZE > $OUT/pmc_fetch_$wl.txt 2>&1
  python $R/tools/rocprof_summary.py pmc $(find $OUT/write_$wl -name "*.db" | head -1) WRITE_SIZE > $OUT/pmc_write_$wl.txt 2>&1
done
declare -A PASS
PASS[B]="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
PASS[C]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
PASS[H]="TCC_HIT_sum TCC_MISS_sum"
for order in 1 0; do
  for p in B C H; do
    FEMCY_PROF_STORAGE_ORDER=$order timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pmcs_$p -o pmc -- python $R/tools/prof_workload.py c3d10 1 0 120 > $OUT/pmcs_$p.log 2>&1
    db=$(find $OUT/pmcs_$p -name "*.db" | head -1)
    if [ -n "$db" ]; then echo "== vectors in storage order: $order" >> $OUT/pmc_spmv_c3d10.txt; python $R/tools/rocprof_summary.py pmc_all $db k_spmv >> $OUT/pmc_spmv_c3d10.txt 2>&1; fi
    rm -rf $OUT/pmcs_$p
  done
done
cd $R
python tools/make_traffic_json.py $HEAD_SHA c3d4:$(find $OUT/fetch_c3d4 -name "*.db" | head -1):$(find $OUT/write_c3d4 -name "*.db" | head -1) c3d10:$(find $OUT/fetch_c3d10 -name "*.db" | head -1):$(find $OUT/write_c3d10 -name "*.db" | head -1) > $OUT/traffic.log 2>&1
cp profiles/spmv_traffic.json $OUT/spmv_traffic.json
rm -rf $OUT/kt_c3d4 $OUT/kt_c3d10 $OUT/fetch_c3d4 $OUT/fetch_c3d10 $OUT/write_c3d4 $OUT/write_c3d10
head -14 $OUT/kernel_stats_c3d4.txt; head -14 $OUT/kernel_stats_c3d10.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
cat $OUT/bench_c3d4.json
ls -la $OUT
