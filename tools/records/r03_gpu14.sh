#!/bin/bash
# SQ / TCP / TA counters of k_assemble_rows4 and k_assemble_rows2 on the C3D10 bench mesh (separate --pmc passes)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03q; mkdir -p $OUT
cd /tmp
declare -A PASS
PASS[A]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
PASS[B]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
PASS[C]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
PASS[D]="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS_ATOMIC SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
for m in 8 6; do
  for p in A B C D; do
    timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/p_${m}_$p -o pmc -- python $R/tools/asm_probe.py c3d10 $m 5 > $OUT/p_${m}_$p.log 2>&1
    db=$(find $OUT/p_${m}_$p -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db k_assemble_rows >> $OUT/pmc_rows4_rows2_sq.txt 2>&1; fi
    rm -rf $OUT/p_${m}_$p
  done
done
cat $OUT/pmc_rows4_rows2_sq.txt
