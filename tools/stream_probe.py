"""stream ceilings of the device by path, launch shape and buffer size (femcy_probe_stream; modes in femcy.h /
kernels_pcg_persist.hip).  usage: python tools/stream_probe.py [sizes MiB ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be

ctx = be.Context(0)
sizes = [int(v) for v in sys.argv[1:]] or [24, 48, 98, 128, 400]
modes = [int(v) for v in os.environ.get("MODES", "0,1,4,5,8,9,10,11,12,13").split(",")]
print("modes: 0 = 1 WG/CU x 8 x 16 B loads in flight per lane, 1 = +nt, 2 / 3 = 8 WG/CU (/ nt), 4 / 5 = 16 / 32 in flight, "
      "6 / 7 = +nt, 8 / 9 / 10 = LDS-DMA 8 / 16 / 32 KiB in flight per wave, 11-13 = +nt")
for mb in sizes:
    row = []
    for mode in modes:
        best = 0.0
        for _ in range(2):
            try:
                g, moved = ctx.probe_stream(mb << 20, 20 if mb < 500 else 8, mode)
            except be.FemcyError as e:
                g = float("nan")
            best = max(best, g)
        row.append(best)
    print(f"  {mb:5d} MiB: " + "  ".join(f"m{k} {v:6.0f}" for k, v in zip(modes, row)), flush=True)
