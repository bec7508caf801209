#!/bin/bash
# small-system PCG with the tagged-granule exchange: probe on three decks + the tests that touch it
cd /root/repo
mkdir -p gpurun_out/r03i
for d in twist_plate_C3D10.inp twist_plate_C3D4.inp ellip_dense_CPS3_0d04.inp; do
  timeout 120 python tools/small_probe.py $d
done > gpurun_out/r03i/small_probe.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_pcg_persist.py tests/test_gpu_parity.py -m gpu -x -q -k "small or pcg" > gpurun_out/r03i/pytest.log 2>&1
tail -n 5 gpurun_out/r03i/pytest.log
cat gpurun_out/r03i/small_probe.txt
