#!/bin/bash
# counters of k_assemble_rows4 with the rows in the caller's numbering and in the measured coordinate order (why is it
# slower there?), and of the current k_pcg_persist
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04h
mkdir -p $OUT
cd /tmp
declare -A PASS
PASS[A]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU"
PASS[B]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"
PASS[C]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
PASS[H]="TCC_HIT_sum TCC_MISS_sum"
PASS[F]="FETCH_SIZE"
PASS[W]="WRITE_SIZE"
for order in 0 1; do
  echo "=========== k_assemble_rows4, FEMCY_OPT_NODE_ORDER = $order" >> $OUT/pmc_rows4_node_order.txt
  for p in A B C H F W; do
    FEMCY_PROBE_NODE_ORDER=$order timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pm_$p -o pmc -- python $R/tools/asm_probe.py c3d10 8 5 > $OUT/pm.log 2>&1
    db=$(find $OUT/pm_$p -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db k_assemble_rows4 >> $OUT/pmc_rows4_node_order.txt 2>&1; fi
    rm -rf $OUT/pm_$p
  done
done
for p in A B C H F W; do
  timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pp_$p -o pmc -- python $R/tools/persist_pmc_driver.py 3 200 > $OUT/pp.log 2>&1
  db=$(find $OUT/pp_$p -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db k_pcg_persist >> $OUT/pmc_persist_c3d4.txt 2>&1; fi
  rm -rf $OUT/pp_$p
done
cat $OUT/pmc_rows4_node_order.txt
cat $OUT/pmc_persist_c3d4.txt
