#!/bin/bash
# round 6, call 8: C3D10 k = 12 (995 328 elements, 4.18 M DOF, 3 GB matrix): kernel trace + FETCH / WRITE of the assembly and
# the product (SURVEY 8d "then raise k"); host memory / cores of the box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06h
mkdir -p $OUT
free -g | head -2; nproc
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $R/tools/r06_c3d10_k12.py 12 > $OUT/k12.json 2> $OUT/k12.err
python $R/tools/rocprof_summary.py stats $(find $OUT/kt -name "*.db" | head -1) > $OUT/r06_kernel_stats_c3d10_k12.txt 2>&1
rm -rf $OUT/kt
for p in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $p -d $OUT/pp -o pmc -- python $R/tools/r06_c3d10_k12.py 12 > /dev/null 2> $OUT/pp.err
  db=$(find $OUT/pp -name "*.db" | head -1)
  for kern in k_assemble_rows4 k_spmv k_geom k_update; do python $R/tools/rocprof_summary.py pmc_all $db $kern >> $OUT/r06_pmc_c3d10_k12.txt 2>&1; done
  rm -rf $OUT/pp
done
head -12 $OUT/r06_kernel_stats_c3d10_k12.txt; cat $OUT/r06_pmc_c3d10_k12.txt
