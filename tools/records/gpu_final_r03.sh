#!/bin/bash
# round-3 record run: full GPU suite, smoke, bench (headline with hbm_bound + cpu_baseline; three-kernel; C3D10; the
# persistent PCG across ranks through a 1-rank RCCL communicator and the RCCL loop beside it), ceilings and variants,
# two ranks on one GPU, kernel traces, PMC traffic passes (bench: FETCH / WRITE -> profiles/spmv_traffic.json;
# rows2 / rows3; the persistent kernel; the C3D10 SpMV)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03final
HEAD_SHA=${1:-unknown}
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1
tail -16 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
FEMCY_BENCH_PERSIST=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --hbm-bound off > $OUT/bench_c3d4_three_kernel.json 2> $OUT/bench_c3d4_three_kernel.err
timeout 300 python bench.py --workload c3d10 --no-cpu-baseline > $OUT/bench_c3d10.json 2> $OUT/bench_c3d10.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm_persistent.json 2> $OUT/bench_forcecomm_persistent.err
FEMCY_BENCH_PERSIST_MULTI=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm_rccl.json 2> $OUT/bench_forcecomm_rccl.err
(timeout 300 python tools/microbench.py 12; timeout 300 python tools/microbench.py 6 1) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $OUT/microbench.txt
FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_allvar.so ITERS=500 timeout 900 python tools/persist_variants.py c3d4 2>&1 | grep -v "amdgpu.ids" > $OUT/persist_variants_c3d4.txt
FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_allvar.so ITERS=300 timeout 900 python tools/persist_variants.py c3d10 0 6 7 2>&1 | grep -E "variant|streamed" > $OUT/persist_variants_c3d10.txt
MODES=0,1,14,16,2,3,9 timeout 300 python tools/stream_probe.py 24 64 98 128 200 400 1024 2>&1 | grep -v amdgpu.ids > $OUT/stream_probe.txt
timeout 300 python tools/multirank_persist_probe.py 300 2>&1 | grep -v amdgpu.ids > $OUT/multirank_persist_probe.txt
(ITERS=500 FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_allvar.so python tools/persist_breakdown.py c3d4 2>&1 | grep "lds") > $OUT/persist_breakdown.txt
cd /tmp
for wl in c3d4 c3d10; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$wl -o kt -- python $R/bench.py --workload $wl --steps 3 --no-cpu-baseline --hbm-bound off --prewarm 1 > $OUT/kt_$wl.log 2>&1
  python $R/tools/rocprof_summary.py stats $(find $OUT/kt_$wl -name "*.db" | head -1) > $OUT/kernel_stats_$wl.txt 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$wl -o pmc -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --hbm-bound off > $OUT/fetch_$wl.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$wl -o pmc -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --hbm-bound off > $OUT/write_$wl.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find $OUT/fetch_$wl -name "*.db" | head -1) FETCH_SIZE > $OUT/pmc_fetch_$wl.txt 2>&1
  python $R/tools/rocprof_summary.py pmc $(find $OUT/write_$wl -name "*.db" | head -1) WRITE_SIZE > $OUT/pmc_write_$wl.txt 2>&1
done
for m in 6 7; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmca_${m}_$ctr -o pmc -- python $R/tools/asm_probe.py c3d10 $m 5 > $OUT/pmca_${m}_$ctr.log 2>&1
    db=$(find $OUT/pmca_${m}_$ctr -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc $db $ctr 2>&1 | grep -E "k_assemble|^kernel" >> $OUT/pmc_rows_c3d10.txt; fi
    rm -rf $OUT/pmca_${m}_$ctr
  done
done
declare -A PASS
PASS[A]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
PASS[B]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
PASS[C]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
PASS[H]="TCC_HIT_sum TCC_MISS_sum"
for p in A B C H; do
  timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pmcp_$p -o pmc -- python $R/tools/persist_pmc_driver.py 3 200 > $OUT/pmcp_$p.log 2>&1
  db=$(find $OUT/pmcp_$p -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db k_pcg_persist >> $OUT/pmc_persist_c3d4.txt 2>&1; fi
  rm -rf $OUT/pmcp_$p
  timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pmcs_$p -o pmc -- python $R/tools/prof_workload.py c3d10 1 20 40 > $OUT/pmcs_$p.log 2>&1
  db=$(find $OUT/pmcs_$p -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db k_spmv >> $OUT/pmc_spmv_c3d10.txt 2>&1; fi
  rm -rf $OUT/pmcs_$p
done
cd $R
python tools/make_traffic_json.py $HEAD_SHA c3d4:$(find $OUT/fetch_c3d4 -name "*.db" | head -1):$(find $OUT/write_c3d4 -name "*.db" | head -1) c3d10:$(find $OUT/fetch_c3d10 -name "*.db" | head -1):$(find $OUT/write_c3d10 -name "*.db" | head -1) > $OUT/traffic.log 2>&1
cp profiles/spmv_traffic.json $OUT/spmv_traffic.json
rm -rf $OUT/kt_c3d4 $OUT/kt_c3d10 $OUT/fetch_c3d4 $OUT/fetch_c3d10 $OUT/write_c3d4 $OUT/write_c3d10
head -14 $OUT/kernel_stats_c3d4.txt; head -14 $OUT/kernel_stats_c3d10.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
cat $OUT/bench_c3d4.json
ls -la $OUT
