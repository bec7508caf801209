"""SURVEY 8d "C3D10, 1 GPU (start; then raise k)": the HBM-bound record of bench.py on the k = K twist plate meshed with
quadratic tetrahedra (k = 12: 995 328 C3D10, 4.18 M DOF, ~3 GB of matrix -- twelve times the Infinity Cache).
usage: python tools/r06_c3d10_k12.py [k=12] -> one JSON record (bench.hbm_bound_record) + set-up times on stderr"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from femcy_amd import backend as be, meshgen  # noqa: E402
from femcy_amd.element_zoo import Element_quadratic_tetrahedral  # noqa: E402
from femcy_amd.material_zoo import LinearIsotropic  # noqa: E402
from femcy_amd.user_defined import user_dirichletBC_values  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
t0 = time.time()
m = meshgen.twist_plate_k(k, quadratic=True)
print(f"[c3d10 k={k}] mesh {m['elements'].shape[0]} elements, {m['nodes'].shape[0]} nodes in {time.time() - t0:.1f} s", file=sys.stderr)
rec = bench.hbm_bound_record(be, f"twist plate C3D10 k={k} ({m['elements'].shape[0]} elements)", m, Element_quadratic_tetrahedral(),
                             LinearIsotropic(*m["elastic"]), user_dirichletBC_values, None, iters=50, spmv_reps=20)
print(json.dumps(rec))
