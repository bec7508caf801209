#!/bin/bash
# round 5, call 7: SQ / TCP / TA / TCC counters of the persistent kernel on its two new configurations (C3D10: matrix
# streamed from HBM; CPE8: 2 x 2 blocks, 8 slices per wave) and, for reference, on the headline mesh; separate --pmc passes
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05g
mkdir -p $OUT
cd /tmp
declare -A PASS
PASS[A]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
PASS[B]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
PASS[C]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
PASS[H]="TCC_HIT_sum TCC_MISS_sum"
for wl in c3d10 cpe8 c3d4; do
  for p in A B C H; do
    timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pp_$p -o pmc -- python $R/tools/persist_pmc_driver.py 3 200 $wl > $OUT/pp.log 2>&1
    db=$(find $OUT/pp_$p -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db k_pcg_persist >> $OUT/pmc_persist_$wl.txt 2>&1; fi
    rm -rf $OUT/pp_$p
  done
  tail -2 $OUT/pp.log
done
ls -la $OUT
