#!/bin/bash
# round 5, call 1: new tests (README digits through the device CG; MULTI kernel changes), the persistent PCG forced on the
# HBM-streamed C3D10 plate (shipped build and the pipelined-stream build), the CG branch through the driver at 116 k DOF,
# the 1-rank-communicator bench (MULTI overhead), the first 2-D bench line
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05a
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "readme or nafems" > $OUT/pytest_readme.log 2>&1; tail -3 $OUT/pytest_readme.log
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_xproc.py tests/test_gpu_pcg_persist.py -q -m gpu -x > $OUT/pytest_multi.log 2>&1; tail -5 $OUT/pytest_multi.log
(timeout 400 python tools/r05_ab.py persist_hbm; FEMCY_HIP_LIB=$R/femcy_amd/libfemcy_hip_pipe.so timeout 400 python tools/r05_ab.py persist_hbm "4:-1:1:0,2:-1:1:0,2:-1:1:16,2:-1:0:16,0:-1:1:0,0:-1:1:16,2:0:1:16") 2>&1 | grep -v "amdgpu.ids\|^+ " > $OUT/persist_hbm.txt
cat $OUT/persist_hbm.txt
SAVE=$OUT/cg_driver_k7.npz timeout 600 python tools/r05_cg_driver.py 7 0.05 2>&1 | grep -v amdgpu.ids > $OUT/cg_driver_k7.txt; cat $OUT/cg_driver_k7.txt
SAVE=$OUT/cg_driver_k7_fine.npz timeout 600 python tools/r05_cg_driver.py 7 0.0125 0.003125 2>&1 | grep -v amdgpu.ids > $OUT/cg_driver_k7_fine.txt; cat $OUT/cg_driver_k7_fine.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm.json 2> $OUT/bench_forcecomm.err; python -c "
import json;d=json.load(open('$OUT/bench_forcecomm.json'));print('forcecomm', d['pcg_us_per_iter'], d['config']['persistent_pcg_across_ranks'])"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off > $OUT/bench_single.json 2> $OUT/bench_single.err; python -c "
import json;d=json.load(open('$OUT/bench_single.json'));print('single', d['pcg_us_per_iter'], d['value'])"
timeout 400 python bench.py --workload cpe8 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cpe8.json 2> $OUT/bench_cpe8.err; tail -3 $OUT/bench_cpe8.err; python -c "
import json;d=json.load(open('$OUT/bench_cpe8.json'));print('cpe8', d['value'], d['pcg_us_per_iter'], d['assembly_ms'], d['roofline'])"
ls -la $OUT
