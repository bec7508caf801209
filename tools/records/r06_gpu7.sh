#!/bin/bash
# round 6, call 7: PMC of the final k_assemble_pairs (default knobs 163) on the CPE8 beam; separate --pmc passes
# pair-list kernel k_assemble_pairs<8,4,2> (mode 9); separate --pmc passes, kernel-trace only
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06e
mkdir -p $OUT
cd /tmp
declare -A PASS
PASS[A]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
PASS[B]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
PASS[C]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
PASS[H]="TCC_HIT_sum TCC_MISS_sum"
PASS[F]="FETCH_SIZE"
PASS[W]="WRITE_SIZE"
PASS[S]="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES"
for mode in 9; do
  name=$([ $mode = 2 ] && echo rows || echo pairs)
  kern=$([ $mode = 2 ] && echo k_assemble_rows || echo k_assemble_pairs)
  for p in A B C H F W S; do
    timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pp_$p -o pmc -- python $R/tools/asm_probe.py cpe8 $mode 10 > $OUT/pp.log 2>&1
    db=$(find $OUT/pp_$p -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db $kern >> $OUT/r06_pmc_asm_cpe8_$name.txt 2>&1; fi
    rm -rf $OUT/pp_$p
  done
  tail -2 $OUT/pp.log
done
cat $OUT/r06_pmc_asm_cpe8_pairs.txt
