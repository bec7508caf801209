#!/bin/bash
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02f
mkdir -p $OUT
cd $R
python - <<'PY' > $OUT/devattr.txt 2>&1
import torch
p = torch.cuda.get_device_properties(0)
print(p)
print("shared_memory_per_block", getattr(p, "shared_memory_per_block", None), "optin", getattr(p, "shared_memory_per_block_optin", None), "per_mp", getattr(p, "shared_memory_per_multiprocessor", None))
PY
cat $OUT/devattr.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pcg or dirichlet or errors" > $OUT/pytest_pcg.log 2>&1
tail -5 $OUT/pytest_pcg.log
timeout 1200 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu --durations=6 > $OUT/pytest_e2e.log 2>&1
tail -14 $OUT/pytest_e2e.log
