"""CPU suite: host-side logic of the product (reader, plugin tables, topology, mesh generator)
and the C-ABI surface (library loads, exports every declared symbol, fails loudly without a GPU)."""
import os
import re

import numpy as np
import pytest

from helpers import ROOT, deck, oracle_material
from femcy_amd import element_zoo as ez
from femcy_amd import material_zoo as mz
from femcy_amd import meshgen
from femcy_amd.body import Body
from femcy_amd.reader import InpInfo
from oracle import femcy_oracle as orc
from oracle.elements import elem_def

PAIRS = {"CPS3": ez.Element_linear_triangular, "CPS4": ez.Element_linear_quadrilateral,
         "CPS6": ez.Element_quadratic_triangular, "CPS8": ez.Element_quadratic_quadrilateral,
         "C3D4": ez.Element_linear_tetrahedral, "C3D10": ez.Element_quadratic_tetrahedral}


# --------------------------------------------------------------------------------- plugins
@pytest.mark.parametrize("etype", list(PAIRS))
def test_element_plugin_matches_oracle_tables(etype):
    """the product's table-driven element classes and the oracle's literal restatement agree."""
    e, d = PAIRS[etype](), elem_def(etype)
    rng = np.random.default_rng(0)
    for _ in range(6):
        c = rng.uniform(-1, 1, d.dm)
        assert np.allclose(e.shapeFunc_pyscope(c), d.N(c), atol=1e-14)
        assert np.allclose(e.dshape_dnat_pyscope(c), d.dN(c), atol=1e-14)
    assert np.array_equal(np.asarray(e.gaussPoints), d.gauss_points)
    assert np.array_equal(np.asarray(e.gaussWeights), d.gauss_weights)
    assert e.integPointNum_eachFacet == d.integPointNum_eachFacet
    assert set(e.facet_natural_coos) == set(d.facet_natural_coos)
    for k in d.facet_natural_coos:
        assert np.allclose(e.facet_natural_coos[k], d.facet_natural_coos[k])
        assert np.allclose(e.facet_point_weights[k], d.facet_point_weights[k])
        assert np.allclose(e.facet_natural_normals[k], d.facet_natural_normals[k])
    assert [tuple(map(tuple, s)) for s in e.inp_surface_num] == [tuple(map(tuple, s)) for s in d.inp_surface_num]
    assert np.allclose(e.extrap_matrix(), d.extrap)
    t = e.tables()
    assert t["dN"].shape == (d.nGP, d.npe, d.dm) and np.allclose(t["dN"], d.dN_table())
    g = rng.standard_normal((d.npe, d.dm))
    assert np.array_equal(e.strainMtrx(g), orc.strain_mtrx(g))
    # reference API surface
    for attr in ("dm", "gaussPoints", "gaussWeights", "integPointNum_eachFacet", "facet_natural_coos",
                 "facet_point_weights", "facet_natural_normals", "inp_surface_num", "shapeFunc", "dshape_dnat",
                 "shapeFunc_pyscope", "dshape_dnat_pyscope", "globalNormal", "strainMtrx", "getMesh", "extrapolate"):
        assert hasattr(e, attr), attr
    assert e.gaussPoints.shape[0] == d.nGP and e.gaussPoints.to_numpy().shape == (d.nGP, d.dm)


def test_global_normal_matches_oracle():
    X = np.array([[0., 0.], [30., 10.], [35., 0.]])      # the reference's own __main__ smoke triangle
    e, d = ez.Element_linear_triangular(), elem_def("CPS3")
    for facet in ([1, 0], [1, 2], [0, 2]):
        n, a = e.globalNormal(X, facet)
        no, ao = orc.global_normal(d, X, facet)
        assert np.allclose(n, no) and np.isclose(a, ao)
        mid = X[facet].mean(axis=0)
        assert n @ (mid - X.mean(axis=0)) > 0 and np.isclose(np.linalg.norm(n), 1.0)      # outward unit normal


def test_material_plugins():
    for cls, kind, p in ((mz.LinearIsotropic, "lin3d", (2e11, .3)), (mz.LinearIsotropicPlaneStrain, "pstrain", (2.1e5, .3)),
                         (mz.LinearIsotropicPlaneStress, "pstress", (2.1e5, .3)), (mz.NeoHookean, "neohooke", (.4, 20.))):
        m, o = cls(*p), orc.Material(kind, p)
        assert np.allclose(m.C, o.C, rtol=1e-15) and m.type == o.type and m.dm == o.dm
        assert np.allclose(m.params, p)
        if hasattr(m, "C_6x6"):
            assert np.allclose(m.C_6x6, o.C_6x6, rtol=1e-15)
        with pytest.raises(TypeError):
            m.constitutiveOfLargeDeform(np.eye(3), None, None)        # device fields only: no CPU path
    assert mz.LinearIsotropicPlaneStrain(1., 0.5).C[0, 0] > 1e29       # nu = 0.5 guard, as the reference


# ---------------------------------------------------------------------------------- reader
def test_reader_on_reference_decks():
    inp = InpInfo(deck("twist_plate_C3D4.inp"))
    assert inp.nodes.shape == (323, 3) and inp.eSets["C3D4"].shape == (1116, 4)
    assert inp.eSets["C3D4"].min() == 0 and inp.eSets["C3D4"].max() == 322
    assert inp.geometric_nonlinear is True
    assert inp.time_incs == {"ini_inc": 0.05, "max_time": 1.0, "min_inc": 1e-5, "max_inc": 0.05}
    assert [(b["dof"], b["user"]) for b in inp.dirichlet_bc_info] == [(0, False), (1, False), (2, False),
                                                                      (0, True), (1, True), (2, True)]
    assert np.allclose(inp.nodes[inp.node_sets["Set-10"]][:, 2], 120.0)
    assert np.allclose(inp.nodes[inp.node_sets["fit_right_z"]][:, 2], 0.0)
    m = inp.materials["Elastic"]
    assert isinstance(m, mz.LinearIsotropic) and m.modulus == 2e11 and m.poisson_ratio == 0.3

    inp = InpInfo(deck("cook_3d_linearEl_largeDef.inp"))
    (key, m), = inp.materials.items()
    assert "neo hooke" in key and isinstance(m, mz.NeoHookean) and m.C1 == 0.4 and m.D1 == 1. / 0.05
    assert len(inp.dirichlet_bc_info) == 4                 # includes the *Boundary block before *Step
    nb, = inp.neumann_bc_info
    assert nb["traction"] == 0.0625 and np.array_equal(nb["direction"], [0., 1., 0.])
    assert all(len(f) == 3 for f in nb["face_set"])

    inp = InpInfo(deck("ellip_membrane_linEle_localVeryFine.inp"))
    nb, = inp.neumann_bc_info
    assert nb["traction"] == 10.0 and "direction" not in nb          # pressure -10 -> traction +10 along the normal
    assert isinstance(list(inp.materials.values())[0], mz.LinearIsotropicPlaneStress)
    assert inp.geometric_nonlinear is False

    inp = InpInfo(deck("beamDeflec_quadPSE_largeD_load800.inp"))      # CPS6: edges split into half-edges
    assert all(len(f) == 2 for f in inp.neumann_bc_info[0]["face_set"])
    assert isinstance(inp.ELE, ez.Element_quadratic_triangular)


def test_reader_quirks(tmp_path):
    base = open(deck("beam_CPS3_disp_meshSize5.inp")).read()
    # substring type match: CPS6M would be read as CPS6; here CPS3 stays CPS3 with junk after it
    p = tmp_path / "a.inp"
    p.write_text(base.replace("type=CPS3", "type=CPS3X"))
    assert list(InpInfo(str(p)).eSets) == ["CPS3"]
    # nlgeom is taken from the LAST comma field of the first *Step line
    p.write_text(base.replace("nlgeom=YES", "nlgeom=NO, inc=100"))
    assert InpInfo(str(p)).geometric_nonlinear is True
    p.write_text(base.replace("nlgeom=YES", "nlgeom=NO"))
    assert InpInfo(str(p)).geometric_nonlinear is False
    # ini_inc is clipped to max_inc
    p.write_text(base.replace("0.25, 1., 1e-05, 0.25", "0.75, 1., 1e-05, 0.25"))
    assert InpInfo(str(p)).time_incs["ini_inc"] == 0.25
    # 2-D decks accept *Elastic only
    p.write_text(base.replace("*Elastic", "*Hyperelastic, neo hooke"))
    with pytest.raises(ValueError):
        InpInfo(str(p))
    # ... unless the plane-strain neo-Hookean extension is asked for, and then only on CPE
    with pytest.raises(ValueError):
        InpInfo(str(p), allow_2d_hyperelastic=True)                   # this deck is CPS3
    p.write_text(base.replace("*Elastic", "*Hyperelastic, neo hooke").replace("type=CPS3", "type=CPE3"))
    m, = InpInfo(str(p), allow_2d_hyperelastic=True).materials.values()
    assert type(m).__name__ == "NeoHookeanPlaneStrain" and m.dm == 2 and m.C.shape == (3, 3)
    # generate expands start, stop, step inclusively
    p.write_text(base.replace("*End Assembly", "*Nset, nset=gen, instance=x, generate\n 3, 9, 3\n*End Assembly"))
    assert InpInfo(str(p)).node_sets["gen"].tolist() == [2, 5, 8]
    # a `**` comment inside a data block does not end it; trailing commas and blank lines are tolerated
    p.write_text(base.replace("*End Assembly", "*Nset, nset=cm, instance=x\n 1, 2, 3,\n** note\n 7, 8\n\n*End Assembly"))
    assert InpInfo(str(p)).node_sets["cm"].tolist() == [0, 1, 2, 6, 7]
    # ... but it does end the *Node block (reference inp_info.py:28-45 breaks at the first line holding a '*')
    lines = base.split("\n")
    k = next(i for i, l in enumerate(lines) if l.startswith("*Node")) + 3
    p.write_text("\n".join(lines[:k] + ["** cut"] + lines[k:]))
    with pytest.raises((IndexError, ValueError)):        # the connectivity now refers to labels that were not read
        InpInfo(str(p))


def test_reader_every_deck_equals_line_by_line_parse():
    """the block-wise numpy parse of nodes / elements equals a plain per-line parse of the same decks."""
    import glob
    for path in sorted(glob.glob(os.path.join(os.path.dirname(deck("twist_plate_C3D4.inp")), "*.inp"))):
        inp = InpInfo(path)
        nodes, conn, mode = {}, [], None
        for line in open(path).read().split("\n"):
            if "*" in line:
                if mode == "n":
                    seen_nodes = True
                mode = "n" if line.startswith("*Node") and not nodes else "e" if line.startswith("*Element") else None
                continue
            if line.strip() and mode == "n":
                t = line.split(",")
                nodes[int(t[0])] = [float(v) for v in t[1:]]
            elif line.strip() and mode == "e":
                conn.extend(int(v) for v in line.rstrip().rstrip(",").split(","))
        order = {lab: i for i, lab in enumerate(nodes)}
        et = list(inp.eSets)[0]
        el = inp.eSets[et]
        ref = np.array(conn).reshape(-1, el.shape[1] + 1)[:, 1:]
        assert np.array_equal(inp.nodes, np.array(list(nodes.values())))
        assert np.array_equal(el, np.vectorize(order.get)(ref))


# ------------------------------------------------------------------------ topology / meshes
def test_body_topology_matches_oracle():
    inp = InpInfo(deck("twist_plate_C3D4.inp"))
    el = inp.eSets["C3D4"]
    body = Body(inp.nodes, el, inp.ELE)
    topo = orc.Topology(inp.nodes, el, elem_def("C3D4"))
    co = body.get_coElement_nodes()
    for a in (0, 17, 322):
        assert co[a] == topo.adj_idx[topo.adj_ptr[a]:topo.adj_ptr[a + 1]].tolist()
        assert body.get_nodeEles()[a] == sorted(np.where((el == a).any(axis=1))[0].tolist())
    assert body.get_boundary() == topo.boundary()
    assert body.get_boundary() == {f: es[0] for f, es in body.facetDic.items() if len(es) == 1}
    ij = topo.sparseIJ()
    assert ij.shape == (969, np.diff(topo.adj_ptr).max() * 3 + 1) and (ij[:, 0] % 3 == 0).all()


def test_png_renderer(tmp_path):
    """headless picture of a result: valid PNG, coloured pixels, 2-D and 3-D meshes."""
    import struct
    import zlib
    from femcy_amd import png_out
    for name in ("ellip_CPS8.inp", "twist_plate_C3D4.inp"):
        inp = InpInfo(deck(name))
        el = list(inp.eSets.values())[0]
        tris = (np.concatenate([el[:, list(t)] for t in inp.ELE._tri_split]) if inp.nodes.shape[1] == 2
                else inp.ELE.getMesh(el)[2])
        path = str(tmp_path / (name + ".png"))
        png_out.render_png(path, inp.nodes, tris, np.linalg.norm(inp.nodes, axis=1), label="r", dpi=50)
        raw = open(path, "rb").read()
        assert raw[:8] == b"\x89PNG\r\n\x1a\n"
        w, h = struct.unpack(">II", raw[16:24])
        assert (w, h) == (400, 300)
        idat = b"".join(raw[m.start() + 4:m.start() + 4 + struct.unpack(">I", raw[m.start() - 4:m.start()])[0]]
                        for m in re.finditer(b"IDAT", raw))
        px = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, -1)[:, 1:].reshape(h, w, -1)[..., :3]
        assert (np.ptp(px.astype(int), axis=2) > 60).mean() > 0.05        # a good share of saturated (jet) pixels
    vals = png_out.nodal_average(np.array([[0, 1, 2], [1, 2, 3]]), np.array([[1., 2., 3.], [4., 5., 6.]]), 5)
    assert np.allclose(vals, [1., 3., 4., 6., 0.])


def test_meshgen_twist_plate():
    m = meshgen.twist_plate_k(1)
    assert m["nodes"].shape == (9 * 2 * 13, 3) and m["elements"].shape == (6 * 8 * 12, 4)
    ed = elem_def("C3D4")
    _, vol = orc.dsdx_and_vol(m["nodes"], m["elements"], np.zeros(m["nodes"].size), ed)
    assert (vol > 0).all() and np.isclose(vol.sum(), 80 * 10 * 120)
    # conforming: every interior face is shared by exactly two tets
    faces = np.sort(np.concatenate([m["elements"][:, list(f)] for f in ((0, 1, 2), (0, 1, 3), (1, 2, 3), (0, 2, 3))]), axis=1)
    _, counts = np.unique(faces, axis=0, return_counts=True)
    assert set(counts.tolist()) == {1, 2} and (counts == 1).sum() == 2 * 2 * (8 * 1 + 1 * 12 + 8 * 12)
    m10 = meshgen.twist_plate_k(1, quadratic=True)
    _, v10 = orc.dsdx_and_vol(m10["nodes"], m10["elements"], np.zeros(m10["nodes"].size), elem_def("C3D10"))
    assert (v10 > 0).all() and np.isclose(v10.sum(), 96000.0)
    assert m10["nodes"].shape[0] == 17 * 3 * 25            # corner + mid-side nodes = the refined grid
    # BASELINE.md sizes
    assert meshgen.scaling_cells(1) == (96, 12, 144) and meshgen.scaling_cells(8) == (192, 24, 288)
    for n in (1, 2, 4, 8):
        nx, ny, nz = meshgen.scaling_cells(n)
        assert nz % n == 0 and 6 * nx * ny * nz // n == 995328


def test_meshgen_inp_roundtrip(tmp_path):
    m = meshgen.twist_plate_k(1)
    p = str(tmp_path / "tw.inp")
    meshgen.write_inp(p, m)
    inp = InpInfo(p)
    assert np.array_equal(inp.nodes, m["nodes"]) and np.array_equal(inp.eSets["C3D4"], m["elements"])
    assert inp.time_incs == m["time_incs"] and inp.geometric_nonlinear
    assert [(b["dof"], b["user"], len(b["node_set"])) for b in inp.dirichlet_bc_info] == \
           [(b["dof"], b["user"], len(b["node_set"])) for b in m["dirichlet_bc_info"]]


# --------------------------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    from femcy_amd import backend as be
    header = open(os.path.join(ROOT, "include", "femcy.h")).read()
    declared = set(re.findall(r"\b(femcy_[a-z_A-Z0-9]+)\s*\(", header))
    assert declared == set(be.EXPORTS), declared ^ set(be.EXPORTS)
    lib = be.load_library(require_gpu_runtime=False)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.femcy_version() >= 100


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from femcy_amd import backend as be
    with pytest.raises(be.FemcyError, match="no HIP device|no CPU path"):
        be.Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "femcy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".sh")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dirpath, f)
                assert "oracle/" not in src.replace("`oracle/`", ""), os.path.join(dirpath, f)


def test_traffic_fingerprint_is_the_kernels_machine_code(tmp_path):
    """bench.kernel_object_sha (round 6): profiles/spmv_traffic.json is pinned to the machine code of every k_pcg_persist /
    k_spmv instantiation inside libfemcy_hip.so, not to whole source files -- an edit elsewhere (round 5: an assembly
    option in ctx.hpp) must not void the PMC record of the driver's line, a changed kernel must.  Checked on the built
    library: the fingerprint is reproducible, covers the PCG kernels only (the assembly kernels hash differently and do
    not enter it), and the committed record -- when it carries one -- is the record of THIS library or is refused loudly."""
    import json
    import bench
    lib = os.path.join(ROOT, "femcy_amd", "libfemcy_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libfemcy_hip.so not built")
    a, b = bench.kernel_object_sha(), bench.kernel_object_sha(lib)
    assert a == b and len(a) == 16
    assert bench.kernel_object_sha(patterns=("k_assemble",)) != a
    assert bench.kernel_object_sha(patterns=("k_spmv",)) != a                # a subset of the kernels: another fingerprint
    with pytest.raises(RuntimeError):
        bench.kernel_object_sha(patterns=("no_such_kernel",))
    # a library whose bytes OUTSIDE those kernels differ has the same fingerprint: flip a byte of the host code
    data = bytearray(open(lib, "rb").read())
    pos = data.find(b"femcy_last_error")                                     # a dynamic-symbol string of the host side
    assert pos > 0
    data[pos + 6] ^= 1
    other = tmp_path / "libother.so"
    other.write_bytes(bytes(data))
    assert bench.kernel_object_sha(str(other)) == a
    # since round 6 each workload of the record is pinned to the ONE instantiation its passes ran (`kernel_sha`): adding or
    # changing OTHER instantiations of the same template does not void it
    assert bench.kernel_symbol_fragment("void femcy::(anonymous namespace)::k_pcg_persist<3, 3, 4, 6, false>(femcy::X)") == \
        "13k_pcg_persistILi3ELi3ELi4ELi6ELb0EE"
    assert bench.kernel_symbol_fragment("void femcy::k_spmv<2, 4, true>(int, femcy::XcdRanges)") == "6k_spmvILi2ELi4ELb1EE"
    one = bench.kernel_object_sha(patterns=("13k_pcg_persistILi3ELi3ELi4ELi6ELb0EE",))
    assert one != a and one != bench.kernel_object_sha(patterns=("13k_pcg_persistILi2ELi8ELi2ELi6ELb0EE",))
    tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    doc = json.load(open(tpath))
    entry = doc["workloads"]["c3d4"]
    val, src = bench.pmc_traffic("c3d4", "k_pcg_persist")
    current = (entry.get("kernel_sha") == bench.kernel_object_sha(patterns=(bench.kernel_symbol_fragment(entry["kernel"]),))
               if entry.get("kernel_sha") else doc.get("kernel_object_sha") == a)
    if current:
        assert val == entry["hbm_bytes_per_launch"] and "spmv_traffic.json" in src
        bad, why = bench.pmc_traffic("c3d4", "k_pcg_persist", layout={"n": 1, "nnzb": 1, "stored_blocks": 1, "nslices": 1})
        assert (bad is None and "layout" in why) or entry.get("layout") is None
    else:
        assert val is None and "stale" in src


def test_compiled_kernels_restore_exec_before_lane_wise_code(tmp_path):
    """guard against a hipcc 7.0 miscompile met in round 6 (k_pcg_persist with 6 / 7 slices of 3 x 3 blocks per wave;
    profiles/r06_persist_spw67_fault.txt): at the join of an exec-masked region the register allocator's copies of
    long-lived values into accumulation registers were placed BEFORE `s_or_b64 exec, exec, sX` -- lanes outside the mask kept
    garbage that was used under full exec much later (a memory fault in the final store of x, or wrong iterates; also, silent
    until then, in the non-default variant <3,4,0,14>).  The source was changed so that the pattern does not arise (masked
    loads -> unconditional loads + selects); this test compiles every HIP translation unit to assembly and checks EVERY
    kernel for lane-wise instructions between such a join and its exec restore (tools/check_exec_joins.py)."""
    import shutil
    import sys
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_exec_joins as cej
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "femcy_amd", "csrc")
    units = ["kernels_pcg_persist", "kernels_pcg", "kernels_assembly", "kernels_direct"]
    procs = [subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-S",
                               "--cuda-device-only", os.path.join(src, u + ".hip"), "-o", str(tmp_path / (u + ".s"))],
                              stderr=subprocess.DEVNULL) for u in units]
    assert all(p.wait() == 0 for p in procs)
    nk = 0
    for u in units:
        for name, lines in cej.kernels((tmp_path / (u + ".s")).read_text()):
            nk += 1
            bad = cej.check(lines)
            assert not bad, (u, name[:100], bad[:2])
    assert nk > 150                                              # every kernel instantiation of the library was looked at
