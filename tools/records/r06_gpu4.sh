#!/bin/bash
# round 6, call 4: FEMCY_TUNE_PAIRS with the Morton chunk order (bit 5) on the CPE8 beam + FETCH/WRITE of the best
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for k in 0 1 32 33 34 35 41 43; do FEMCY_PROBE_PAIRS=$k python tools/asm_probe.py cpe8 9 30 2>&1 | grep "mode 9"; done | tee gpurun_out/r06_asm_cpe8_knobs2.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06d; mkdir -p $OUT; cd /tmp
for k in 33 35; do for p in FETCH_SIZE WRITE_SIZE; do
  FEMCY_PROBE_PAIRS=$k timeout 300 rocprofv3 --kernel-trace --pmc $p -d $OUT/pp -o pmc -- python $R/tools/asm_probe.py cpe8 9 10 > $OUT/pp.log 2>&1
  db=$(find $OUT/pp -name "*.db" | head -1); echo "knobs $k" >> $OUT/fetch.txt; python $R/tools/rocprof_summary.py pmc_all $db k_assemble_pairs >> $OUT/fetch.txt 2>&1; rm -rf $OUT/pp
done; done; cat $OUT/fetch.txt
