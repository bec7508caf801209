cd /root/repo
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c3d10 or twist_plate_1M" 2>&1 | tail -n 3
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pins.py -m gpu -x -q 2>&1 | tail -n 3
timeout 200 python tools/microbench.py 6 1 2>&1 | grep -i "assemble\|geom"
