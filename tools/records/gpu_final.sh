#!/bin/bash
# round-2 record run: full GPU suite, smoke, both bench workloads, kernel traces, PMC traffic passes, 8 M on one GPU
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02final
HEAD_SHA=${1:-unknown}
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1
tail -20 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
cat $OUT/bench_c3d4.json
FEMCY_BENCH_PERSIST=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3d4_three_kernel.json 2> $OUT/bench_c3d4_three_kernel.err
cat $OUT/bench_c3d4_three_kernel.json
timeout 300 python bench.py --workload c3d10 > $OUT/bench_c3d10.json 2> $OUT/bench_c3d10.err
cat $OUT/bench_c3d10.json
timeout 600 python bench.py --no-cpu-baseline --steps 3 --cells 192,24,288 --prewarm 1 > $OUT/bench_8M.json 2> $OUT/bench_8M.err
cat $OUT/bench_8M.json
(timeout 300 python tools/microbench.py 12; timeout 300 python tools/microbench.py 6 1) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $OUT/microbench.txt
(ITERS=500 python tools/persist_breakdown.py c3d4 2>&1 | grep "lds") > $OUT/persist_breakdown.txt
(python tools/small_probe.py twist_plate_C3D10.inp; python tools/small_probe.py twist_plate_C3D4.inp; python tools/small_probe.py ellip_dense_CPS3_0d04.inp) 2>&1 | grep "iteration\|n =" > $OUT/small_probe.txt
cat $OUT/small_probe.txt
cd /tmp
for wl in c3d4 c3d10; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$wl -o kt -- python $R/bench.py --workload $wl --steps 3 --no-cpu-baseline --prewarm 1 > $OUT/kt_$wl.log 2>&1
  python $R/tools/rocprof_summary.py stats $(find $OUT/kt_$wl -name "*.db" | head -1) > $OUT/kernel_stats_$wl.txt 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$wl -o pmc -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline > $OUT/fetch_$wl.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$wl -o pmc -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline > $OUT/write_$wl.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find $OUT/fetch_$wl -name "*.db" | head -1) FETCH_SIZE > $OUT/pmc_fetch_$wl.txt 2>&1
  python $R/tools/rocprof_summary.py pmc $(find $OUT/write_$wl -name "*.db" | head -1) WRITE_SIZE > $OUT/pmc_write_$wl.txt 2>&1
done
declare -A PASS
PASS[A]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
PASS[B]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
PASS[C]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
PASS[D]="FETCH_SIZE"
PASS[E]="WRITE_SIZE"
PASS[G]="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_LDS_ATOMIC SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
for p in A B C D E G; do
  timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pmcr_$p -o pmc -- python $R/tools/asm_probe.py c3d10 6 5 > $OUT/pmcr_$p.log 2>&1
  db=$(find $OUT/pmcr_$p -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db k_assemble_rows2 > $OUT/pmc_rows2_$p.txt 2>&1; fi
  rm -rf $OUT/pmcr_$p
done
for p in A B C G; do
  timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pmcp_$p -o pmc -- python $R/tools/persist_pmc_driver.py 3 200 > $OUT/pmcp_$p.log 2>&1
  db=$(find $OUT/pmcp_$p -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db k_pcg_persist > $OUT/pmc_persist_$p.txt 2>&1; fi
  rm -rf $OUT/pmcp_$p
done
cd $R
python tools/make_traffic_json.py $HEAD_SHA c3d4:$(find $OUT/fetch_c3d4 -name "*.db" | head -1):$(find $OUT/write_c3d4 -name "*.db" | head -1) c3d10:$(find $OUT/fetch_c3d10 -name "*.db" | head -1):$(find $OUT/write_c3d10 -name "*.db" | head -1) > $OUT/traffic.log 2>&1
cp profiles/spmv_traffic.json $OUT/spmv_traffic.json
rm -rf $OUT/kt_c3d4 $OUT/kt_c3d10 $OUT/fetch_c3d4 $OUT/fetch_c3d10 $OUT/write_c3d4 $OUT/write_c3d10
head -16 $OUT/kernel_stats_c3d4.txt; head -16 $OUT/kernel_stats_c3d10.txt
timeout 300 python bench.py --steps 5 --no-cpu-baseline --prewarm 1 > $OUT/bench_c3d4_with_traffic.json 2>/dev/null
cat $OUT/bench_c3d4_with_traffic.json
ls -la $OUT
