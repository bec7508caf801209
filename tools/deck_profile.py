"""where the wall time of a whole deck run goes (host Python vs device): cProfile of femcy_amd.main on one deck
usage: python tools/deck_profile.py [deck=twist_plate_C3D10.inp]"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from femcy_amd import main as fmain

name = sys.argv[1] if len(sys.argv) > 1 else "twist_plate_C3D10.inp"
path = os.path.join(ROOT, "tests", "golden", "decks", name)
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
fmain.main([path, "--quiet"])
pr.disable()
print(f"wall {time.perf_counter() - t:.2f} s")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
