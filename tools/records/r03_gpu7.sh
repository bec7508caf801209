#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03g
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -x -q -m gpu > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench_c3d4.json 2> $OUT/bench_c3d4.err
python - <<PY
import json
d=json.load(open('$OUT/bench_c3d4.json'))
print('value', d['value'], 'us/it', d['pcg_us_per_iter'])
for r in d['hbm_bound']:
    print(r['workload'][:40], 'spmv us', r['spmv']['avg_launch_us'], 'frac', r['spmv']['frac'], 'iter us', r['pcg_iteration']['us'], 'frac', r['pcg_iteration']['frac'])
c=d['cpu_baseline']; print('cpu', c['value'], c['cores'], c['gbs'], c['threads_scan_iters_per_s'], c['container_cpu_quota'], c['numa_nodes'], c.get('host_backend'))
PY
FEMCY_BENCH_PERSIST=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --hbm-bound off > $OUT/bench_c3d4_3k.json 2> $OUT/bench_c3d4_3k.err
python -c "import json;d=json.load(open('$OUT/bench_c3d4_3k.json'));print('three-kernel', d['value'],d['pcg_us_per_iter'],d['roofline']['avg_launch_us'],d['roofline']['frac'])"
timeout 600 python bench.py --workload c3d10 --no-cpu-baseline > $OUT/bench_c3d10.json 2> $OUT/bench_c3d10.err
python -c "import json;d=json.load(open('$OUT/bench_c3d10.json'));print('c3d10', d['value'],d['pcg_us_per_iter'],d['roofline']['avg_launch_us'],d['roofline']['frac'], d['assembly_ms'])"
