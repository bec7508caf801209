#!/bin/bash
# rows4 as the C3D10 default: full GPU suite, C3D10 bench, microbench, FETCH / WRITE of the assembly kernels
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03k
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -n 3 $OUT/pytest_gpu.log
timeout 300 python bench.py --workload c3d10 --no-cpu-baseline > $OUT/bench_c3d10.json 2> $OUT/bench_c3d10.err
timeout 300 python tools/microbench.py 6 1 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $OUT/microbench_c3d10.txt
cd /tmp
for m in 8; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmca_${m}_$ctr -o pmc -- python $R/tools/asm_probe.py c3d10 $m 5 > $OUT/pmca_${m}_$ctr.log 2>&1
    db=$(find $OUT/pmca_${m}_$ctr -name "*.db" | head -1)
    if [ -n "$db" ]; then python $R/tools/rocprof_summary.py pmc $db $ctr 2>&1 | grep -E "k_assemble|^kernel" >> $OUT/pmc_rows4_c3d10.txt; fi
    rm -rf $OUT/pmca_${m}_$ctr
  done
done
cat $OUT/pmc_rows4_c3d10.txt; grep -i "assemble\|geom" $OUT/microbench_c3d10.txt; tail -c 400 $OUT/bench_c3d10.json
