#!/bin/bash
# round-2 batch 1: new full-size tests, bench (both workloads), micro-benchmarks, PMC passes of the C3D10 kernels
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu --durations=8 > $OUT/pytest_fullsize.log 2>&1
tail -15 $OUT/pytest_fullsize.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_bench_contract.py -x -q -m gpu > $OUT/pytest_parity.log 2>&1
tail -5 $OUT/pytest_parity.log
timeout 300 python bench.py > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
cat $OUT/bench_c3d4.json
timeout 300 python bench.py --workload c3d10 > $OUT/bench_c3d10.json 2> $OUT/bench_c3d10.err
cat $OUT/bench_c3d10.json
(timeout 300 python tools/microbench.py 12; timeout 300 python tools/microbench.py 6 1) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $OUT/microbench.txt
cat $OUT/microbench.txt
cd /tmp
declare -A PASS
PASS[A]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
PASS[B]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
PASS[C]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
PASS[D]="FETCH_SIZE"
PASS[E]="WRITE_SIZE"
PASS[F]="TCC_HIT_sum TCC_MISS_sum"
PASS[G]="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_LDS_ATOMIC SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
for p in A B C D E F G; do
  timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pmc_$p -o pmc -- python $R/tools/prof_workload.py c3d10 > $OUT/pmc_$p.log 2>&1
  db=$(find $OUT/pmc_$p -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db > $OUT/pmc_c3d10_$p.txt 2>&1; fi
  rm -rf $OUT/pmc_$p
done
grep -A12 "k_assemble_rows\|k_spmv" $OUT/pmc_c3d10_A.txt | head -60
ls -la $OUT
