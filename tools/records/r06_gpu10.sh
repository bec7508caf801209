#!/bin/bash
# round 6, call 10: FETCH / WRITE of k_assemble_rows4 on the k = 12 C3D10 plate under both launch orders
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06j
mkdir -p $OUT
cd /tmp
for o in 2; do for p in FETCH_SIZE WRITE_SIZE; do
  FEMCY_PROBE_K=12 FEMCY_PROBE_ROWS4_ORDER=$o timeout 300 rocprofv3 --kernel-trace --pmc $p -d $OUT/pp -o pmc -- python $R/tools/asm_probe.py c3d10 8 5 > $OUT/pp.log 2>&1
  db=$(find $OUT/pp -name "*.db" | head -1); echo "launch order $o" >> $OUT/r06_pmc_rows4_order_k12.txt; python $R/tools/rocprof_summary.py pmc_all $db k_assemble_rows4 >> $OUT/r06_pmc_rows4_order_k12.txt 2>&1; rm -rf $OUT/pp
done; done
cat $OUT/r06_pmc_rows4_order_k12.txt
