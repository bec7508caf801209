"""CPU suite: libfemcy_cpu.so, the host implementation of include/femcy.h (femcy_amd/csrc_cpu/, SURVEY.md 8b "a CPU
implementation of the same ABI", BASELINE configs[0] "plumbing, no GPU").

The parity tests of the HIP path are written against `femcy_amd.backend.Context`, i.e. against the C ABI -- so the same
files, unchanged, are run here in a child process with FEMCY_BACKEND=cpu: element matrices, forces, Dirichlet / Neumann
treatment, PCG iterates, post-processing, golden vectors, first-principles pins, the consistent tangent, and whole
decks through the reference's increment / Newton driver, all against the oracle.  (Device-only machinery is skipped by
tests/conftest.py; the fine C3D10 twist deck -- minutes of host factorisations -- is left to the GPU suite.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, deck


def _run(files, k=None, timeout=1500):
    env = dict(os.environ, FEMCY_BACKEND="cpu")
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + [os.path.join(ROOT, "tests", f) for f in files]
    if k:
        cmd += ["-k", k]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1500:]
    assert " passed" in tail and "failed" not in tail, tail
    return tail


def test_library_builds_and_exports_every_declared_symbol():
    from femcy_amd import backend as be
    if not os.path.exists(be.CPU_LIB_PATH):
        subprocess.check_call(["bash", os.path.join(ROOT, "femcy_amd", "csrc_cpu", "build.sh")])
    lib = be.load_library(kind="cpu")
    missing = [s for s in be.EXPORTS if not hasattr(lib, s)]
    assert not missing, missing


def test_cpu_backend_is_explicit_opt_in_only():
    """the default backend is the HIP library; nothing falls back to the host by itself"""
    from femcy_amd import backend as be
    env = {k: v for k, v in os.environ.items() if k != "FEMCY_BACKEND"}
    code = ("from femcy_amd import backend as be\n"
            "assert be.default_backend() == 'hip'\n"
            "import torch\n"
            "if not torch.cuda.is_available():\n"
            "    try:\n"
            "        be.Context(0)\n"
            "    except be.FemcyError as e:\n"
            "        assert 'no HIP device' in str(e) or 'not found' in str(e), str(e)\n"
            "    else:\n"
            "        raise SystemExit('a Context was created without a GPU and without FEMCY_BACKEND=cpu')\n"
            "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr[-1500:]
    with pytest.raises(be.FemcyError):
        os.environ["FEMCY_BACKEND"] = "tpu"
        try:
            be.default_backend()
        finally:
            del os.environ["FEMCY_BACKEND"]


def test_cpu_context_in_process_solves_a_deck_system():
    """Context(backend='cpu') next to the default backend in one process: K, f and a PCG solve on the C3D4 twist deck
    against the oracle; results do not depend on the OpenMP thread count (fixed-chunk reductions, owner-computes rows)"""
    from femcy_amd import backend as be
    from femcy_amd.reader import InpInfo
    from helpers import oracle_material
    from oracle import femcy_oracle as orc
    from oracle.elements import elem_def
    inp = InpInfo(deck("twist_plate_C3D4.inp"))
    et = list(inp.eSets)[0]
    el = inp.eSets[et]
    mat = list(inp.materials.values())[0]
    ctx = be.Context(0, backend="cpu")
    ctx.set_mesh(inp.nodes, el)
    ctx.set_element(inp.ELE)
    ctx.set_material(mat)
    info = ctx.build_pattern()
    assert info.stored_blocks == info.nnzb                             # block-CSR: no padding
    u = 0.01 * np.sin(np.arange(ctx.n) * 0.3)
    ctx.upload(be.VEC_DOF, u)
    ctx.assemble_K(be.VEC_DOF)
    topo = orc.Topology(inp.nodes, el, elem_def(et))
    Ko = orc.assemble_K(topo, u, oracle_material(mat).C)
    K = ctx.get_K_bsr().tocsr()
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in inp.dirichlet_bc_info]))
    b = np.cos(np.arange(ctx.n) * 0.7)
    ctx.upload(be.VEC_RESIDUAL, b)
    ctx.dirichlet_newton(cons, be.VEC_RESIDUAL)
    bb = ctx.download(be.VEC_RESIDUAL)
    Kd = orc._zero_rows_cols_unit_diag(Ko, cons)
    res = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=30)
    x = ctx.download(be.VEC_X)
    xo, ito, r0o, rmo = orc.pcg_reference(Kd, bb, eps=0.0, maxit=30)
    assert res[0] == ito == 30 and res[1] == r0o and abs(res[2] - rmo) <= 1e-10 * rmo
    assert np.linalg.norm(x - xo) <= 1e-10 * np.linalg.norm(xo)
    ctx.close()
    # thread-count independence, bit for bit (child processes: OMP_NUM_THREADS is read when the runtime starts)
    code = ("import numpy as np, sys\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from femcy_amd import backend as be\n"
            "from femcy_amd.reader import InpInfo\n"
            "from helpers import deck\n"
            "inp = InpInfo(deck('twist_plate_C3D4.inp')); el = list(inp.eSets.values())[0]\n"
            "c = be.Context(0, backend='cpu'); c.set_mesh(inp.nodes, el); c.set_element(inp.ELE)\n"
            "c.set_material(list(inp.materials.values())[0]); c.build_pattern()\n"
            "c.upload(be.VEC_DOF, 0.01 * np.sin(np.arange(c.n) * 0.3)); c.residual_and_K(be.VEC_DOF, be.VEC_RESIDUAL)\n"
            "cons = np.unique(np.concatenate([np.asarray(b['node_set']) * 3 + b['dof'] for b in inp.dirichlet_bc_info]))\n"
            "c.dirichlet_newton(cons, be.VEC_RESIDUAL); r = c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=40)\n"
            "import hashlib; print(hashlib.sha256(c.download(be.VEC_X).tobytes()).hexdigest(), r[2].hex())\n"
            % (ROOT, os.path.join(ROOT, "tests")))
    outs = set()
    for nt in ("1", "3", "8"):
        o = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT,
                           env=dict(os.environ, OMP_NUM_THREADS=nt))
        assert o.returncode == 0, o.stderr[-1500:]
        outs.add(o.stdout.strip().splitlines()[-1])
    assert len(outs) == 1, outs


def test_parity_suites_of_the_c_abi_on_the_cpu_backend():
    tail = _run(["test_gpu_parity.py", "test_gpu_pins.py", "test_gpu_tangent.py", "test_gpu_neohooke2d.py",
                 "test_gpu_direct.py"])
    print("[cpu backend] parity / pins / tangent / neo-Hookean 2-D / direct solve:", tail)


def test_deck_parity_on_the_cpu_backend():
    """whole decks through System_of_equations.solve (increments, modified Newton, line searches, cut-backs) on the
    host backend against the oracle's displacements: every deck of the reference but the fine C3D10 twist deck (2 923
    factorisations of 9 558 unknowns: 6 minutes of host time; the coarse one is in since the direct branch is a
    factorisation)"""
    tail = _run(["test_gpu_e2e.py"], k="not twist_plate_C3D10")
    print("[cpu backend] decks:", tail)


def test_cg_branch_of_the_driver_on_the_cpu_backend():
    """the reference's >= 1e5-DOF leg (`solve_by_CG`: eps = 1e-3, maxit = n) through the driver on the host backend: the
    116 k-DOF twist plate in sound increments (12 CG solves, the oracle's iteration counts, displacements to 1e-6) and
    the 108 k-DOF linear CPE8 cantilever (one solve).  (The deck's own first increment -- 820 k CG iterations, most of
    them solves that run to the reference's cap on an indefinite K -- is the GPU suite's: 11 s there, 6 minutes here.)"""
    tail = _run(["test_gpu_cg_branch.py"], k="beam_lin or fine_per_solve")
    assert "2 passed" in tail
    print("[cpu backend] CG branch:", tail)
    # ... and the reference's own decks on that leg (cg_branch_from = 0): the single-solve ones here, all 48 in the GPU suite
    tail = _run(["test_gpu_cg_branch.py"], k="cg_leg and (ellip_ or cook_3d or smallD or smallDef)")
    print("[cpu backend] CG leg on the decks:", tail)
