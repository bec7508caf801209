#!/bin/bash
# round-5 record run, last part (after the direct-solve changes: refinement threshold, matrix-core update): the GPU suite,
# smoke, the default line
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05final
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
python -c "
import json;d=json.load(open('$OUT/bench_c3d4.json'));print(d['value'], d['pcg_us_per_iter'], d['roofline']['traffic'], [(x.get('dof'), round(x.get('direct_ms',0),2), round(x.get('tight_pcg_ms',0),2)) for x in d['direct_branch']])"
