#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02d
mkdir -p $OUT
cd $R
for v in NOATOMIC NOCOMPUTE NOWRITE; do
  FEMCY_HIP_LIB=$R/build/libfemcy_$v.so timeout 120 python tools/asm_probe.py c3d10 2>&1 | grep "mode" >> $OUT/probe.txt
done
timeout 120 python tools/asm_probe.py c3d10 2>&1 | grep "mode" >> $OUT/probe.txt
timeout 120 python tools/asm_probe.py c3d10 2 2>&1 | grep "mode" >> $OUT/probe.txt
cat $OUT/probe.txt
cd /tmp
declare -A PASS
PASS[A]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
PASS[B]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
PASS[C]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
PASS[E]="WRITE_SIZE"
PASS[G]="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_LDS_ATOMIC SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
for p in A B C E G; do
  timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pmc_$p -o pmc -- python $R/tools/asm_probe.py c3d10 6 5 > $OUT/pmc_$p.log 2>&1
  db=$(find $OUT/pmc_$p -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db k_assemble_rows2 > $OUT/pmc_rows2_$p.txt 2>&1; fi
  rm -rf $OUT/pmc_$p
done
cat $OUT/pmc_rows2_*.txt
