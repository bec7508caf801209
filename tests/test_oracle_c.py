"""CPU suite: the as-written C/OpenMP restatement (oracle/femcy_oracle.c, the timed CPU baseline)
against the numpy oracle, function by function."""
import numpy as np
import pytest

from helpers import deck, oracle_material
from femcy_amd.reader import InpInfo
from oracle import femcy_oracle as orc
from oracle.c_oracle import COracle
from oracle.elements import elem_def

DECKS = ["ellip_membrane_linEle_localVeryFine.inp", "cookMembrane_2d_linearEl_smallDef.inp", "ellip_CPS4.inp",
         "ellip_membrane_quadritic_trig_neumann.inp", "ellip_CPS8.inp", "twist_plate_C3D4.inp",
         "cook_3d_linearEl_largeDef.inp", "twist_C3D10_coarse.inp"]
KIND = {"lin3d": 0, "pstrain": 1, "pstress": 2, "neohooke": 3}


def rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("name", DECKS)
def test_c_oracle_matches_numpy_oracle(name):
    inp = InpInfo(deck(name))
    et = list(inp.eSets)[0]
    el = inp.eSets[et]
    ed = elem_def(et)
    mat = oracle_material(list(inp.materials.values())[0])
    topo = orc.Topology(inp.nodes, el, ed)
    co = COracle(inp.nodes, el, ed.dN_table(), ed.gauss_weights, mat.C, topo.adj_ptr, topo.adj_idx)
    assert np.array_equal(co.ij, topo.sparseIJ())
    L = np.ptp(inp.nodes, axis=0).max()
    u = 0.02 * L * np.sin(np.arange(topo.n) * 0.13)
    co.get_dsdx_and_vol(u)
    dsdx, vol = orc.dsdx_and_vol(topo.nodes, topo.elements, u, ed)
    assert rel(co.dsdx, dsdx) < 1e-13 and rel(co.vol, vol) < 1e-13
    co.assemble()
    K = orc.assemble_K(topo, u, mat.C)
    assert abs(co.to_csr() - K).max() / abs(K).max() < 1e-13
    f = co.internal_force(u, KIND[mat.kind], *mat.params)
    fo, sig, F, _, _ = orc.internal_force(topo, u, mat)
    assert rel(co.F, F) < 1e-13 and rel(co.sigma, sig) < 1e-11 and rel(f, fo) < 1e-11
    x = np.cos(np.arange(topo.n) * 0.7)
    assert rel(co.compute_Ad(x), K @ x) < 1e-13


def test_c_oracle_cg_is_the_reference_recurrence():
    inp = InpInfo(deck("twist_plate_C3D4.inp"))
    el = inp.eSets["C3D4"]
    ed = elem_def("C3D4")
    mat = oracle_material(list(inp.materials.values())[0])
    topo = orc.Topology(inp.nodes, el, ed)
    co = COracle(inp.nodes, el, ed.dN_table(), ed.gauss_weights, mat.C, topo.adj_ptr, topo.adj_idx)
    co.get_dsdx_and_vol(np.zeros(topo.n))
    co.assemble()
    cons = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in inp.dirichlet_bc_info]))
    co.zero_rows_cols_unit_diag(cons)
    K = co.to_csr()
    b = np.sin(np.arange(topo.n) * 0.11) * 1e3
    b[cons] = 0.0
    for eps in (1e-3, 1e-8):
        x, it, r0, rmax = co.cg(b, eps=eps)
        xo, ito, r0o, rmaxo = orc.pcg_reference(K, b, eps=eps)
        assert abs(it - ito) <= 2 and r0 == r0o                 # OpenMP reduction order can move the stop by one
        if it == ito:
            assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-5    # reduction-order rounding x cond(K)
        assert np.abs(K @ x - b).max() < 2 * eps * r0
    x5, it5, _, _ = co.cg(b, eps=0.0, maxit=5)
    assert it5 == 5
