#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04dbg
timeout 600 python -m pytest tests/test_gpu_pcg_persist.py -q -m gpu -k "variants_agree" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_pcg_persist.py -q -m gpu -k "barrier_timeout_falls_back" 2>&1 | tail -15
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import numpy as np, sys
sys.path.insert(0,'.')
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic
m = meshgen.twist_plate(24, 6, 96)
ctx = be.Context(0)
ctx.set_mesh(m["nodes"], m["elements"]); ctx.set_element(Element_linear_tetrahedral()); ctx.set_material(LinearIsotropic(*m["elastic"]))
info = ctx.build_pattern(); ctx.assemble_K(-1)
cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3); ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
def paths():
    t = ctx.timing(); return t["solves_three"], t["solves_small"], t["solves_persist"], t["barrier_timeouts"]
ctx.set_option(be.OPT_PCG_PERSIST, 2)
for var in (0, 6, 14):
    ctx.set_option(be.TUNE_PERSIST_VARIANT, var)
    for eps, maxit in ((0.0, 25), (1e-9, 10**6), (1e-9, 5000), (0.0, 25)):
        b = paths(); r = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=eps, maxit=maxit); a = paths()
        print("variant", var, "eps", eps, "maxit", maxit, "->", r, "paths delta", tuple(x - y for x, y in zip(a, b)), flush=True)
ctx.set_option(be.TUNE_PERSIST_VARIANT, -1)
ctx.set_option(107, 512)
b = paths(); r = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=12); a = paths()
print("wgs 512 ->", r, "paths delta", tuple(x - y for x, y in zip(a, b)), flush=True)
PY
