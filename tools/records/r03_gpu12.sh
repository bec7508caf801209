#!/bin/bash
# after k_assemble_rows4 became the C3D10 default: kernel stats of the C3D10 bench, both micro-benchmarks, headline bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03l
mkdir -p $OUT
cd $R
(timeout 300 python tools/microbench.py 12; timeout 300 python tools/microbench.py 6 1) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $OUT/microbench.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_c3d10 -o kt -- python $R/bench.py --workload c3d10 --steps 3 --no-cpu-baseline --hbm-bound off --prewarm 1 > $OUT/kt_c3d10.log 2>&1
python $R/tools/rocprof_summary.py stats $(find $OUT/kt_c3d10 -name "*.db" | head -1) > $OUT/kernel_stats_c3d10.txt 2>&1
rm -rf $OUT/kt_c3d10
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
head -n 8 $OUT/kernel_stats_c3d10.txt; tail -c 300 $OUT/bench_c3d4.json
