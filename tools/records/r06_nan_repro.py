"""minimal reproduction of the round-5 NaN: one in-process multi-rank test, then the three-launch PCG on the C3D10 plate"""
import os, sys
import numpy as np
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); os.chdir(ROOT)
pre = sys.argv[1] if len(sys.argv) > 1 else "tests/test_gpu_multirank.py::test_neighbour_exchange_equals_allreduce"
if pre != "none":
    print("pre rc", pytest.main(["-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", pre]), flush=True)
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic
first = os.environ.get("FIRST", "c3d4")
def system(m, ele):
    ctx = be.Context(0)
    ctx.set_mesh(m["nodes"], m["elements"]); ctx.set_element(ele); ctx.set_material(LinearIsotropic(*m["elastic"]))
    info = ctx.build_pattern(); ctx.assemble_K(-1)
    cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
    ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(ctx.n) * 0.11) * 1e3)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    return ctx, info
if first == "c3d4":
    ctx, info = system(meshgen.twist_plate(100, 12, 152), Element_linear_tetrahedral())
    for persist in (0, 1):
        ctx.set_option(be.OPT_PCG_PERSIST, persist)
        print("c3d4 persist", persist, ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=9), flush=True)
    ctx.close()
ctx, info = system(meshgen.twist_plate(48, 6, 72, quadratic=True), Element_quadratic_tetrahedral())
ctx.set_option(be.TUNE_PERSIST_MAX_MB, int(os.environ.get("MAXMB", "240")))
try:
    print("c3d10", ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=20), ctx.timing()["solves_three"], flush=True)
except be.FemcyError as e:
    print("FAILED:", e, flush=True)
    K = ctx.get_K_bsr()
    print("K finite", np.isfinite(K.data).all(), "b finite", np.isfinite(ctx.download(be.VEC_RESIDUAL)).all())
    Kc = K.tocsr(); dg = Kc.diagonal(); rowabs = np.asarray(abs(Kc).sum(axis=1)).ravel()
    zr = np.nonzero(rowabs == 0)[0]
    print("zero diagonal entries", int((dg == 0).sum()), "all-zero rows", zr.size, "first/last zero-row nodes", (zr[:3] // 3, zr[-3:] // 3) if zr.size else None, "of", ctx.n // 3)
    for mode in (be.ASM_ROWS4, be.ASM_ROWS2, be.ASM_GATHER):
        ctx.set_option(be.OPT_ASSEMBLY, mode); ctx.assemble_K(-1)
        Kc = ctx.get_K_bsr().tocsr(); rowabs = np.asarray(abs(Kc).sum(axis=1)).ravel()
        print("  re-assembled with mode", mode, ": all-zero rows", int((rowabs == 0).sum()))
    ctx.set_option(be.OPT_ASSEMBLY, be.ASM_AUTO)
    x = np.ones(ctx.n); ctx.upload(be.VEC_TMP0, x); ctx.spmv(be.VEC_TMP0, be.VEC_TMP1); y = ctx.download(be.VEC_TMP1)
    print("public spmv finite", np.isfinite(y).all(), "vs scipy", np.abs(y - K @ x).max() / np.abs(y).max())
    def attempt_with(label, opt, val, back):
        ctx.set_option(opt, val)
        try:
            print(f"  {label}: OK", ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=20), flush=True)
        except be.FemcyError as e3:
            print(f"  {label}: FAILED ({str(e3)[-60:]})", flush=True)
        ctx.set_option(opt, back)
    attempt_with("node-order vectors (OPT_PCG_STORAGE_ORDER 0)", be.OPT_PCG_STORAGE_ORDER, 0, 1)
    for w in (1, 2, 4):
        attempt_with(f"spmv waves per slice {w}", be.OPT_SPMV_VARIANT, w, 0)
    attempt_with("matrix stream temporal (TUNE_SPMV_NT 0)", 102, 0, -1)
    attempt_with("matrix stream nt (TUNE_SPMV_NT 1)", 102, 1, -1)
    attempt_with("vectors nt (103 = 1)", 103, 1, -1)
    attempt_with("keep permille 0 (110)", 110, 0, -1)
    attempt_with("keep permille 1000 (110)", 110, 1000, -1)
    attempt_with("graph off", be.OPT_PCG_GRAPH, 0, 1)
    attempt_with("poll 1", be.OPT_PCG_POLL, 1, 32)
    attempt_with("fused update", be.OPT_PCG_FUSED_UPDATE, 1, 0)
    for attempt in range(2):
        try:
            print("retry", attempt, ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=20), flush=True)
        except be.FemcyError as e2:
            print("retry", attempt, "FAILED again:", e2, flush=True)
ctx.close()
