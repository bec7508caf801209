#!/bin/bash
# round 5, call 16: counters of k_assemble_rows4, shipped against FEMCY_ROWS4_TILE=4,19 and 2,28 (separate --pmc passes),
# and the kernel-trace time of each
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05p
mkdir -p $OUT
cd /tmp
declare -A PASS
PASS[C]="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_TA_BUSY_sum SQ_BUSY_CYCLES"
PASS[F]="FETCH_SIZE"
PASS[W]="WRITE_SIZE"
for cfg in "" "4,19" "2,28"; do
  echo "=========== k_assemble_rows4, FEMCY_ROWS4_TILE = [$cfg]" >> $OUT/pmc_rows4_tile.txt
  FEMCY_ROWS4_TILE=$cfg timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $R/tools/asm_probe.py c3d10 8 30 > $OUT/kt.log 2>&1
  f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Name|k_assemble_rows4" $f | cut -c1-300 >> $OUT/pmc_rows4_tile.txt; fi
  rm -rf $OUT/kt
  for p in C F W; do
    FEMCY_ROWS4_TILE=$cfg timeout 300 rocprofv3 --kernel-trace --pmc ${PASS[$p]} -d $OUT/pm_$p -o pmc -- python $R/tools/asm_probe.py c3d10 8 5 > $OUT/pm.log 2>&1
    db=$(find $OUT/pm_$p -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc_all $db k_assemble_rows4 >> $OUT/pmc_rows4_tile.txt 2>&1; fi
    rm -rf $OUT/pm_$p
  done
done
cat $OUT/pmc_rows4_tile.txt
