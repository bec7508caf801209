"""round 4 A/B runs on one GPU, one process per library build:
  persist   the persistent PCG on the 1 M C3D4 plate: variant 6 (three exchanges) against 14 (two exchanges, in-band
            validity of the published d), register rows 4 / 5; iterates compared with variant 6
  order     the three-launch PCG on the C3D10 plate and the 8 M C3D4 plate: vectors in node / storage order
            (FEMCY_OPT_PCG_STORAGE_ORDER) x rows ordered by the caller's numbering / the measured coordinate order
            (FEMCY_OPT_NODE_ORDER): SpMV and PCG iteration (HIP events), geometry + assembly, gather cost of the
            pattern, solution compared with the first configuration
usage: [FEMCY_HIP_LIB=...] python tools/r04_ab.py persist|order [c3d10|c3d4_8m|c3d4]"""
import os
import sys
import time

# numpy's BLAS pool spinning after a norm() eats the container's CPU quota and stalls the host for 35-75 ms now and then
# (visible as one slow repetition in three in the first round-4 records): one thread is enough here
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femcy_amd import backend as be, meshgen
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
from femcy_amd.material_zoo import LinearIsotropic
from femcy_amd.user_defined import user_dirichletBC_values

HBM = 8000.0


def problem(wl):
    quad = wl == "c3d10"
    m = meshgen.twist_plate(48, 6, 72, quadratic=True) if quad else meshgen.twist_plate_k(24 if wl == "c3d4_8m" else 12)
    u = np.zeros(m["nodes"].size)
    cons = []
    for bc in m["dirichlet_bc_info"]:
        cons.append(np.asarray(bc["node_set"]) * 3 + bc["dof"])
        if bc["user"]:
            user_dirichletBC_values(u, bc["node_set"], 3, bc["dof"], m["nodes"], 0.05)
    return m, quad, u, np.unique(np.concatenate(cons)).astype(np.int32)


def make_ctx(m, quad, opts):
    ctx = be.Context(0)
    for k, v in opts:
        ctx.set_option(k, v)
    ctx.set_mesh(m["nodes"], m["elements"])
    ctx.set_element(Element_quadratic_tetrahedral() if quad else Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    return ctx, ctx.build_pattern()


def state(ctx, u, cons):
    ctx.upload(be.VEC_DOF, u)
    ctx.vector(be.VEC_RHS).fill(0.0)
    ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
    ctx.vec_sub(be.VEC_RESIDUAL, be.VEC_FORCE, be.VEC_RHS)
    ctx.assemble_K(be.VEC_DOF)
    cs = ctx.dofset(cons)
    ctx.dofset_dirichlet_newton(cs, be.VEC_RESIDUAL)
    return cs


def persist(wl="c3d4"):
    m, quad, u, cons = problem(wl)
    nit = int(os.environ.get("ITERS", "500"))
    ctx, info = make_ctx(m, quad, [])
    state(ctx, u, cons)
    print(f"lib {os.path.basename(be.LIB_PATH)}  {wl}: n {ctx.n}, streamed {ctx.persist_streamed_bytes() / 1e6:.1f} MB / iteration", flush=True)
    ref = None
    combos = ((6, 4), (6, 4), (6, 4)) if os.environ.get("ONLY6") else ((6, 4), (14, 4), (14, 5), (6, 4), (14, 4))
    for var, rj in combos:
        try:
            ctx.set_option(be.TUNE_PERSIST_VARIANT, var)
            ctx.set_option(105, rj)
            ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=nit)
            times = []
            for _ in range(5):
                ctx.sync()
                t = time.perf_counter()
                it, r0, rmax = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=nit)
                times.append((time.perf_counter() - t) / nit * 1e6)
            x = ctx.download(be.VEC_X)
            it30 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
            x30 = ctx.download(be.VEC_X)
            if ref is None:
                ref = (x30.copy(), it30)
            d30 = np.linalg.norm(x30 - ref[0]) / np.linalg.norm(ref[0])
            again = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
            same = np.array_equal(ctx.download(be.VEC_X), x30)
            conv = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
            tm = ctx.timing()
            print(f"  variant {var:2d} rj {rj}: " + " ".join(f"{t:6.2f}" for t in times) + f" us/it | 30 its rmax {it30[2]:.9e} "
                  f"|x-x6|/|x6| {d30:.1e} reproducible {same} | eps 1e-3: {conv[0]} its | persist/three/timeouts "
                  f"{tm['solves_persist']}/{tm['solves_three']}/{tm['barrier_timeouts']}", flush=True)
        except be.FemcyError as e:
            print(f"  variant {var} rj {rj}: FAILED {e}", flush=True)
    ctx.close()


def order(wl):
    m, quad, u, cons = problem(wl)
    ref = None
    for node_order, pos in ((0, 0), (0, 1), (1, 1), (1, 0)):
        ctx, info = make_ctx(m, quad, [(be.OPT_NODE_ORDER, node_order), (be.OPT_PCG_STORAGE_ORDER, pos), (be.OPT_PCG_PERSIST, 0)])
        used, lines = ctx.node_order()
        state(ctx, u, cons)
        nnz, nnzb = info.nnz, info.nnzb
        spmv_b = 8 * nnz + 4 * nnzb + 4 * (ctx.nn + 1) + 16 * ctx.n
        iter_b = spmv_b + 88 * ctx.n
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=100)
        ctx.set_option(be.OPT_TIMING, 8)
        ctx.timing_reset()
        its = 0
        for _ in range(3):
            ctx.assemble_K(be.VEC_DOF)
            ctx.dofset_dirichlet_newton(ctx.dofset(cons), be.VEC_RESIDUAL)
            its += ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=200)[0]
        tm = ctx.timing()
        ctx.set_option(be.OPT_TIMING, 0)
        spmv_us = tm["spmv_ms"] * 1e3 / max(tm["spmv_launches"], 1)
        iter_us = tm["pcg_ms"] * 1e3 / its
        asm_ms = (tm["geom_ms"] + tm["assemble_ms"]) / max(tm["assemble_launches"], 1)
        r30 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
        x30 = ctx.download(be.VEC_X)
        # the product through the public entry point (node order in, node order out) must not depend on the row order
        ctx.upload(be.VEC_TMP0, np.sin(np.arange(ctx.n) * 0.37))
        ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
        y = ctx.download(be.VEC_TMP1)
        if ref is None:
            ref = (x30.copy(), y.copy())
        dx = np.linalg.norm(x30 - ref[0]) / np.linalg.norm(ref[0])
        dy = np.abs(y - ref[1]).max() / np.abs(ref[1]).max()
        print(f"  {wl} node_order {node_order} (used {used}; lines/gather " + " ".join(f"{v:.1f}" for v in lines if v) +
              f") storage-order vectors {pos}: SpMV {spmv_us:7.2f} us = {spmv_b / spmv_us / 1e3 / HBM:.3f} of HBM | PCG "
              f"{iter_us:7.2f} us/it = {iter_b / iter_us / 1e3 / HBM:.3f} | geometry + assembly {asm_ms:.3f} ms | padding "
              f"{info.stored_blocks / nnzb - 1:.4f} | 30 its: {r30[0]} rmax {r30[2]:.6e} |x-x0|/|x0| {dx:.1e} |Kx-Kx0| {dy:.1e}",
              flush=True)
        ctx.close()


def knobs(wl):
    """launch-shape knobs of the storage-order product inside the PCG: wavefronts per slice x workgroups per XCD x share of
    the matrix kept on the default cache policy (per mille; -1 = the 235 MB rule)"""
    m, quad, u, cons = problem(wl)
    ctx, info = make_ctx(m, quad, [(be.OPT_PCG_PERSIST, 0)])
    cs = state(ctx, u, cons)
    spmv_b = 8 * info.nnz + 4 * info.nnzb + 4 * (ctx.nn + 1) + 16 * ctx.n
    iter_b = spmv_b + 88 * ctx.n
    combos = [(0, 256, -1), (4, 512, -1), (4, 1024 // 2, 500), (2, 256, -1), (2, 512, -1), (1, 512, -1), (4, 256, 400), (4, 256, 800),
              (0, 256, -1)]
    for wps, cap, keep in combos:
        try:
            ctx.set_option(be.OPT_SPMV_VARIANT, wps)
            ctx.set_option(101, cap)
            ctx.set_option(110, keep)
            ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=100)
            res = []
            for _ in range(2):
                ctx.set_option(be.OPT_TIMING, 4)
                ctx.timing_reset()
                its = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=200)[0]
                tm = ctx.timing()
                spmv_us = tm["spmv_ms"] * 1e3 / max(tm["spmv_launches"], 1)
                ctx.set_option(be.OPT_TIMING, 64)
                ctx.timing_reset()
                its = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=300)[0]
                tm = ctx.timing()
                ctx.set_option(be.OPT_TIMING, 0)
                res.append((spmv_us, tm["pcg_ms"] * 1e3 / its))
            print(f"  {wl} wps {wps} wg/xcd {cap} keep {keep}: " + "  ".join(f"SpMV {a:6.2f} us ({spmv_b / a / 1e3 / HBM:.3f}) PCG {b:6.2f} us/it ({iter_b / b / 1e3 / HBM:.3f})" for a, b in res), flush=True)
        except be.FemcyError as e:
            print(f"  wps {wps} cap {cap} keep {keep}: FAILED {e}", flush=True)
    ctx.close()


def persist_order(cells="40,5,60"):
    """the persistent PCG on a C3D10 plate small enough for it, rows in the caller's numbering / in the measured coordinate
    order (its d is gathered in storage order: 27 against 14 cache lines per gather)"""
    nx, ny, nz = (int(v) for v in cells.split(","))
    m = meshgen.twist_plate(nx, ny, nz, quadratic=True)
    u = np.zeros(m["nodes"].size)
    cons = []
    for bc in m["dirichlet_bc_info"]:
        cons.append(np.asarray(bc["node_set"]) * 3 + bc["dof"])
        if bc["user"]:
            user_dirichletBC_values(u, bc["node_set"], 3, bc["dof"], m["nodes"], 0.05)
    cons = np.unique(np.concatenate(cons)).astype(np.int32)
    ref = None
    for order in (0, 1, 0, 1):
        ctx, info = make_ctx(m, True, [(be.OPT_NODE_ORDER, order)])
        used, lines = ctx.node_order()
        state(ctx, u, cons)
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=200)
        res = []
        for _ in range(4):
            ctx.set_option(be.OPT_TIMING, 1)
            ctx.timing_reset()
            its = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=400)[0]
            tm = ctx.timing()
            ctx.set_option(be.OPT_TIMING, 0)
            res.append(tm["persist_ms"] * 1e3 / max(its, 1) if tm["persist_launches"] else tm["pcg_ms"] * 1e3 / its)
        r30 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
        x30 = ctx.download(be.VEC_X)
        if ref is None:
            ref = x30.copy()
        tm = ctx.timing()
        asm = []
        ctx.set_option(be.OPT_TIMING, 1)
        ctx.timing_reset()
        for _ in range(5):
            ctx.assemble_K(be.VEC_DOF)
        t2 = ctx.timing()
        ctx.set_option(be.OPT_TIMING, 0)
        print(f"  C3D10 {cells} ({ctx.ne} elements, {ctx.nn} nodes, {info.nslices} slices, streamed {ctx.persist_streamed_bytes() / 1e6:.0f} MB) node_order "
              f"{order} (used {used}; lines " + " ".join(f"{v:.1f}" for v in lines if v) + "): persistent PCG " + " ".join(f"{v:6.2f}" for v in res) +
              f" us/it | paths persist/three {tm['solves_persist']}/{tm['solves_three']} | assembly "
              f"{(t2['geom_ms'] + t2['assemble_ms']) / t2['assemble_launches']:.3f} ms | |x-x0|/|x0| {np.linalg.norm(x30 - ref) / np.linalg.norm(ref):.1e}", flush=True)
        ctx.close()


def footprint(wl):
    """the storage-order product with gathers from global memory against the footprint product (FEMCY_OPT_SPMV_FOOTPRINT)"""
    m, quad, u, cons = problem(wl)
    ctx, info = make_ctx(m, quad, [(be.OPT_PCG_PERSIST, 0)])
    state(ctx, u, cons)
    spmv_b = 8 * info.nnz + 4 * info.nnzb + 4 * (ctx.nn + 1) + 16 * ctx.n
    iter_b = spmv_b + 88 * ctx.n
    ref = None
    for flag in (0, 1, 0, 1):
        ctx.set_option(be.OPT_SPMV_FOOTPRINT, flag)
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=100)
        b2b = min(ctx.probe_spmv(200, True) for _ in range(3))
        res = []
        for _ in range(3):
            ctx.set_option(be.OPT_TIMING, 1 << 20)
            ctx.timing_reset()
            its = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=300)[0]
            tm = ctx.timing()
            ctx.set_option(be.OPT_TIMING, 0)
            res.append(tm["pcg_ms"] * 1e3 / its)
        r30 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
        x30 = ctx.download(be.VEC_X)
        conv = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
        if ref is None:
            ref = x30.copy()
        print(f"  {wl} footprint {flag}: SpMV launch to launch {b2b:7.2f} us ({spmv_b / b2b / 1e3 / HBM:.3f} of HBM) | PCG " +
              " ".join(f"{v:7.2f}" for v in res) + f" us/it ({iter_b / min(res) / 1e3 / HBM:.3f}) | 30 its rmax {r30[2]:.9e} |x-x0|/|x0| "
              f"{np.linalg.norm(x30 - ref) / np.linalg.norm(ref):.1e} | eps 1e-3: {conv[0]} its", flush=True)
    ctx.close()


def fused(wl):
    """the single-rank three-launch loop with two vector kernels per iteration against ONE (FEMCY_OPT_PCG_FUSED_UPDATE)"""
    m, quad, u, cons = problem(wl)
    ctx, info = make_ctx(m, quad, [(be.OPT_PCG_PERSIST, 0)])
    state(ctx, u, cons)
    iter_b = 8 * info.nnz + 4 * info.nnzb + 4 * (ctx.nn + 1) + 16 * ctx.n + 88 * ctx.n
    ref = None
    for flag in (0, 1, 0, 1):
        ctx.set_option(be.OPT_PCG_FUSED_UPDATE, flag)
        ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=100)
        res = []
        for _ in range(3):
            ctx.set_option(be.OPT_TIMING, 1 << 20)
            ctx.timing_reset()
            its = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=300)[0]
            tm = ctx.timing()
            ctx.set_option(be.OPT_TIMING, 0)
            res.append(tm["pcg_ms"] * 1e3 / its)
        r30 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
        x30 = ctx.download(be.VEC_X)
        conv = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-3)
        if ref is None:
            ref = x30.copy()
        tm = ctx.timing()
        print(f"  {wl} fused {flag}: PCG " + " ".join(f"{v:7.2f}" for v in res) + f" us/it ({iter_b / min(res) / 1e3 / HBM:.3f} of HBM) | 30 its rmax "
              f"{r30[2]:.9e} |x-x0|/|x0| {np.linalg.norm(x30 - ref) / np.linalg.norm(ref):.1e} | eps 1e-3: {conv[0]} its | three/timeouts "
              f"{tm['solves_three']}/{tm['barrier_timeouts']}", flush=True)
    ctx.close()


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "persist":
        persist(*(sys.argv[2:3]))
    elif what == "persist_order":
        persist_order(*(sys.argv[2:3]))
    elif what == "footprint":
        footprint(sys.argv[2] if len(sys.argv) > 2 else "c3d10")
    elif what == "fused":
        fused(sys.argv[2] if len(sys.argv) > 2 else "c3d10")
    elif what == "knobs":
        knobs(sys.argv[2] if len(sys.argv) > 2 else "c3d10")
    else:
        order(sys.argv[2] if len(sys.argv) > 2 else "c3d10")
