"""The one unexplained NaN of round 5 (DESIGN.md "open items"): the three-launch PCG on the 124 k C3D10 plate returned "NaN
after 1 iteration" once, inside test_persistent_pcg_four_slices_per_wave, in a process that had run the multi-rank files
before it.  This driver repeats that order of events in ONE process:

  phase A (N times): tests/test_gpu_multirank.py (in-process ranks, co-dependent persistent kernels, time-out test) followed
                     by test_persistent_pcg_four_slices_per_wave -- pytest.main in the same interpreter each time;
  phase B (M times): context churn -- create a context on the C3D10 plate, assemble, one three-launch solve (the call that
                     failed), one persistent solve, destroy; every third round with a second live context of another mesh
                     whose buffers are freed in between (address reuse across contexts).

usage: [FEMCY_DEBUG_POISON=1] python tools/r06_nan_hunt.py [N=10] [M=50]   -> a line per round, non-zero exit on any failure"""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bad = 0
t0 = time.time()
for k in range(N):
    rc = pytest.main(["-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_gpu_multirank.py",
                      "tests/test_gpu_pcg_persist.py::test_persistent_pcg_four_slices_per_wave"])
    print(f"[nan hunt] phase A round {k + 1}/{N}: pytest rc {int(rc)}  ({time.time() - t0:.0f} s)", flush=True)
    bad += int(rc) != 0

from femcy_amd import backend as be, meshgen  # noqa: E402
from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral  # noqa: E402
from femcy_amd.material_zoo import LinearIsotropic  # noqa: E402

mq = meshgen.twist_plate(48, 6, 72, quadratic=True)
ml = meshgen.twist_plate(40, 6, 60)
consq = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in mq["dirichlet_bc_info"]]))
bq = np.sin(np.arange(mq["nodes"].size) * 0.11) * 1e3
ref = None
for k in range(M):
    other = None
    if k % 3 == 0:                                                # a second context whose buffers come and go
        other = be.Context(0)
        other.set_mesh(ml["nodes"], ml["elements"])
        other.set_element(Element_linear_tetrahedral())
        other.set_material(LinearIsotropic(*ml["elastic"]))
        other.build_pattern()
        other.assemble_K(-1)
    ctx = be.Context(0)
    try:
        ctx.set_mesh(mq["nodes"], mq["elements"])
        ctx.set_element(Element_quadratic_tetrahedral())
        ctx.set_material(LinearIsotropic(*mq["elastic"]))
        ctx.build_pattern()
        ctx.assemble_K(-1)
        ctx.upload(be.VEC_RESIDUAL, bq)
        ctx.dirichlet_newton(consq, be.VEC_RESIDUAL)
        if other is not None:
            other.close()                                         # frees its buffers while ctx is alive
            other = None
        ctx.set_option(be.TUNE_PERSIST_MAX_MB, 240)               # the rule that sends this matrix to three launches
        r3 = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=20)
        x3 = ctx.download(be.VEC_X)
        ctx.set_option(be.TUNE_PERSIST_MAX_MB, 0)
        rp = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=20)
        xp = ctx.download(be.VEC_X)
        ok = np.isfinite(x3).all() and np.isfinite(xp).all() and r3[0] == rp[0] == 20
        if ref is None:
            ref = (r3, x3.copy(), rp, xp.copy())
        ok = ok and r3 == ref[0] and np.array_equal(x3, ref[1]) and rp == ref[2] and np.array_equal(xp, ref[3])
        print(f"[nan hunt] phase B round {k + 1}/{M}: three-launch {r3}, persistent {rp}, same bits as round 1: {bool(ok)}", flush=True)
        bad += not ok
    except be.FemcyError as e:
        print(f"[nan hunt] phase B round {k + 1}/{M}: FemcyError {e}", flush=True)
        bad += 1
    finally:
        ctx.close()
        if other is not None:
            other.close()
print(f"[nan hunt] {N} + {M} rounds in {time.time() - t0:.0f} s: {'CLEAN' if bad == 0 else str(bad) + ' FAILURES'}")
sys.exit(1 if bad else 0)
