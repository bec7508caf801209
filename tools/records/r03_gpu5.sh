#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_bench_contract.py tests/test_gpu_pcg_persist.py -x -q -m gpu > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --hbm-bound off > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
python -c "import json;d=json.load(open('$OUT/bench_c3d4.json'));print(d['value'],d['pcg_us_per_iter'],d['roofline']['frac'],d['roofline'].get('time_model'))"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm.json 2> $OUT/bench_forcecomm.err
python -c "import json;d=json.load(open('$OUT/bench_forcecomm.json'));print(d['value'],d['pcg_us_per_iter'],d['config'])"
tail -3 $OUT/bench_forcecomm.err
FEMCY_BENCH_PERSIST_MULTI=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off --force-comm > $OUT/bench_forcecomm_rccl.json 2> $OUT/bench_forcecomm_rccl.err
python -c "import json;d=json.load(open('$OUT/bench_forcecomm_rccl.json'));print(d['value'],d['pcg_us_per_iter'],d['config'])"
