"""Generate the committed golden vectors under tests/golden/ from the CPU oracle.

The reference itself cannot run here (Taichi is not installable), so these are NOT
outputs of the reference: they are outputs of the oracle restatement, which is pinned against the
reference's published known answers (tests/test_oracle_pins.py).  They freeze the oracle's results
so that (a) the -m gpu suite can compare the HIP path without re-running 40 s Newton solves on the
CPU and (b) any later change to the oracle that moves a result is caught by the CPU suite.

    python tests/golden/make_golden.py            # solves the decks that are not in oracle_solutions.npz yet
    python tests/golden/make_golden.py --all      # re-solves everything (about 85 minutes, 71 of them the C3D10 twist)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from helpers import deck, oracle_system_from_inp, oracle_material  # noqa: E402
from femcy_amd.reader import InpInfo  # noqa: E402
from oracle import femcy_oracle as orc  # noqa: E402
from oracle.elements import elem_def  # noqa: E402

SOLVE_DECKS = [
    "ellip_membrane_linEle_localVeryFine.inp", "ellip_membrane_quadritic_trig_neumann.inp", "ellip_CPS4.inp",
    "ellip_CPS8.inp", "ellip_membrane_3d_linearEl.inp", "ellip_membrane_3d.inp",
    "ellip_membrane_localFine_dirichlet.inp", "ellip_localVeryFine_directional_force.inp",
    "cookMembrane_2d_linearEl_smallDef.inp", "beam_CPS3_disp_meshSize5.inp", "cook_3d_linearEl_largeDef.inp",
    "beamDeflec_quadPSE_largeD_load800.inp", "twist_plate_C3D4.inp", "twist_C3D10_coarse.inp",
    "cookMembrane_2d_linearEl_largeDef.inp",
    # second batch of reference decks (tests/golden/decks/SOURCES.md): quadratic plane-strain triangles (CPE6), large
    # deformation at two load levels, nu = 0.4999, a free-end traction beam, the fixX beam variant, the densest
    # linear CPS6 deck (29 252 DOF) and a small-deformation C3D10 deck with a surface load
    "cookMembrane_CPE6_largeDef.inp", "cookMembrane_CPE6_largeDef_5MPa.inp", "cookMembrane_CPE6_smallDef_nu0d4999.inp",
    "beamFreeDeflect_CPS6_load_mesh4.inp", "beamDeflec_quadPSE_largeD_load800_fixX.inp", "ellip_dense_CPS6_0d04.inp",
    "cook_3d_quadEl_smallDef.inp",
    # third batch: the rest of the reference's small decks -- mesh-size series of the displacement- and traction-driven
    # beams (three of them end with "allowable minimum dt is reached": the failure path is part of the behaviour),
    # small-deformation variants, nu = 0.4999 with linear triangles, CPE6 at 3.5 MPa
    "beamFreeDeflect_CPS6_load_mesh13.inp", "beamFreeDeflect_CPS6_load_mesh10.inp",
    "beam_CPS6_disp_meshSize10.inp", "beamFreeDeflect_CPS6_load_mesh8.inp", "beam_CPS6_disp_meshSize8.inp",
    "beamFreeDeflect_CPS3_load_mesh5.inp", "beamFreeDeflect_CPS3_load_mesh4.inp", "beam_CPS3_disp_meshSize4.inp",
    "beam_CPS6_disp_meshSize4.inp", "beamFreeDeflect_CPS3_load_mesh2.inp", "beam_CPS3_disp_meshSize2.inp",
    "beamFreeDeflect_CPS6_load_mesh2.inp", "beam_CPS6_disp_meshSize2.inp", "beamFreeDeflect_CPS3_load_mesh1.inp",
    "beam_CPS3_disp_meshSize1.inp", "cook_3d_linearEl_smallDef.inp",
    "cookMembrane_2d_linearEl_smallDef_nu0d4999.inp", "beamDeflec_quadPSE_smallD_load800_fixX.inp",
    "beamDeflec_quadPSE_smallD_load100_fixX.inp", "beamDeflec_quadPSE_smallD_load800_freeEnd.inp",
    "cookMembrane_CPE6_smallDef.inp", "cookMembrane_CPE6_smallDef_3d5MPa.inp",
    "cookMembrane_CPE6_largeDef_3d5MPa.inp",
    "ellip_dense_CPS3_0d04.inp",      # densest CPS3 deck (NAFEMS LE1 convergence series of the README)
    "twist_plate_C3D10.inp",          # the full C3D10 twist: 225 increments, 2923 solves -- 71 minutes of numpy oracle
    # generated decks (femcy_amd.meshgen.beam_quad8, written by write_generated_decks() below):
    # BASELINE configs[1] asks for a CPE8 large-deformation beam, which the reference does not ship
    "gen_beam_CPE8_tip4.inp",       # plane strain StVK, 20 x 2 quad8, converges in 4 increments
    "gen_beam_CPS8_tip8.inp",       # plane stress, 3 increment cut-backs (dt/4 + dof_old restore path)
]


def write_generated_decks():
    from femcy_amd import meshgen
    meshgen.write_inp(deck("gen_beam_CPE8_tip4.inp"), meshgen.beam_quad8(20, 2, plane="CPE8", tip_disp=4.0))
    meshgen.write_inp(deck("gen_beam_CPS8_tip8.inp"), meshgen.beam_quad8(20, 2, plane="CPS8", tip_disp=8.0))


def main():
    write_generated_decks()
    out = {}
    path = os.path.join(HERE, "oracle_solutions.npz")
    if "--all" not in sys.argv and os.path.exists(path):
        old = np.load(path)
        out = {k: old[k] for k in old.files}
    for name in SOLVE_DECKS:
        if name[:-4] + "/dof" in out:
            continue
        inp = InpInfo(deck(name))
        s = oracle_system_from_inp(inp)
        t = time.time()
        u = s.solve(inp.time_incs, inp.dirichlet_bc_info, inp.neumann_bc_info)
        key = name[:-4]
        out[key + "/dof"] = u
        out[key + "/meta"] = np.array([len(s.increments), s.n_solves, s.n_assemblies,
                                       getattr(s, "ini_residual", 0.0), float(s.time0)], dtype=np.float64)
        print(f"{name}: |u|={np.linalg.norm(u):.10g} incs={len(s.increments)} solves={s.n_solves} "
              f"({time.time()-t:.1f}s)")
    np.savez_compressed(os.path.join(HERE, "oracle_solutions.npz"), **out)

    # element-level vectors: one Ke per element type (first element of a deck, u = 0)
    ke = {}
    for name in ["ellip_membrane_linEle_localVeryFine.inp", "ellip_CPS4.inp",
                 "ellip_membrane_quadritic_trig_neumann.inp", "ellip_CPS8.inp", "twist_plate_C3D4.inp",
                 "twist_C3D10_coarse.inp"]:
        inp = InpInfo(deck(name))
        et = list(inp.eSets)[0]
        el = inp.eSets[et][:1]
        ed = elem_def(et)
        mat = oracle_material(list(inp.materials.values())[0])
        dsdx, vol = orc.dsdx_and_vol(inp.nodes, el, np.zeros(inp.nodes.size), ed)
        ke[et] = orc.element_stiffness(dsdx, vol, mat.C)[0]
        ke[et + "/dsdx"] = dsdx[0]
        ke[et + "/vol"] = vol[0]
    np.savez_compressed(os.path.join(HERE, "oracle_element_vectors.npz"), **ke)


if __name__ == "__main__":
    main()
