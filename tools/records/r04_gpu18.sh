#!/bin/bash
# round 4, groundwork for the next attempt at k_assemble_rows4: what saturates when a third workgroup per CU is added?
# (-DFEMCY_ROWS4_FAKE_LMAX=40 / 28: accumulators sized for shorter rows = 3 / 4 workgroups per CU, WRONG results, timing
#  only; on top of it the compile-time switches of profiles/r03_rows4_probe.txt remove one piece of the step at a time)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04rows4
mkdir -p $OUT
{
echo "shipped (2 workgroups per CU):"; timeout 200 python tools/asm_probe.py c3d10 8 20 2>&1 | grep assemble
for v in occ3 occ4 occ3_noatomic occ3_noldsread occ3_norecords occ3_nostores occ3_nostaging; do
  echo "$v:"; FEMCY_HIP_LIB=$GRAFT_REPO_ROOT/femcy_amd/probe_libs/libfemcy_$v.so timeout 200 python tools/asm_probe.py c3d10 8 20 2>&1 | grep assemble
done
} | tee $OUT/rows4_occupancy_probe.txt
