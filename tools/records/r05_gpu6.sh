#!/bin/bash
# round 5, call 6: the C3D10 CG-branch test against its new fixture; the cpe8 line with the corrected PMC entry
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05f
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_cg_branch.py -q -m gpu -s -k "c3d10" > $OUT/pytest_cg_c3d10.log 2>&1; tail -8 $OUT/pytest_cg_c3d10.log
timeout 400 python bench.py --workload cpe8 --steps 10 --no-cpu-baseline > $OUT/bench_cpe8.json 2> $OUT/bench_cpe8.err; python -c "
import json;d=json.load(open('$OUT/bench_cpe8.json'));r=d['roofline'];print('cpe8', d['value'], d['pcg_us_per_iter'], r['traffic'], r.get('traffic_over_moved'), r['traffic_source'])"
