"""-m gpu: the multi-rank device path on ONE GPU.  N contexts, one host thread each, joined by the in-process
group transport (femcy_comm_local_id): sub-assembled K per rank, interface pack / all-reduce / unpack, owner
masks in the dot products, the two collectives per CG iteration -- exactly the kernels and call sequence the RCCL
transport runs with one process per GPU (bench.py --gpus N), checked against the single-context solve of the
un-partitioned mesh and against the oracle."""
import threading

import numpy as np
import pytest

from helpers import deck, oracle_material
from oracle import femcy_oracle as orc
from oracle.elements import elem_def

pytestmark = pytest.mark.gpu


def run_ranks(nranks, fn):
    """fn(rank) on one thread per rank; re-raises the first failure."""
    out, err = [None] * nranks, []

    def work(r):
        try:
            out[r] = fn(r)
        except BaseException as e:                      # noqa: BLE001 - reported below
            err.append((r, e))

    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in threads), "a rank is still running"
    if err:
        raise err[0][1]
    return out


def setup_rank(be, part, inp, mat, uid):
    ctx = be.Context(0)
    ctx.set_mesh(part.nodes, part.elements)
    ctx.set_element(inp.ELE)
    ctx.set_material(mat)
    ctx.build_pattern()
    ctx.comm_init(part.rank, part.nranks, uid, part.iface_local_dofs, part.iface_global_slot, part.niface_global,
                  part.owner)
    return ctx


@pytest.mark.parametrize("name,nranks", [("twist_plate_C3D4.inp", 2), ("twist_plate_C3D4.inp", 4),
                                         ("twist_C3D10_coarse.inp", 3), ("ellip_CPS8.inp", 2)])
def test_partitioned_solve_equals_single_context(gpu_ctx_factory, name, nranks):
    from femcy_amd import backend as be, partition
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck(name))
    et = list(inp.eSets)[0]
    el = inp.eSets[et]
    mat = list(inp.materials.values())[0]
    dm = inp.nodes.shape[1]
    n = inp.nodes.shape[0] * dm
    axis = 2 if dm == 3 else 0
    parts = partition.build_all_parts(inp.nodes, el, nranks, axis=axis)
    assert all(p.niface_global > 0 for p in parts)

    def smooth(nodes):
        L = np.ptp(inp.nodes, axis=0).max()
        x = nodes / L
        return (0.02 * L * np.stack([np.sin(1.3 * x[:, 0] + 0.4), 0.5 * np.cos(2.1 * x[:, 1] - 0.2),
                                     0.3 * np.sin(x.sum(axis=1))][:dm], axis=1)).ravel()

    u_g = smooth(inp.nodes)
    cons_nodes = [(np.asarray(b["node_set"]), b["dof"]) for b in inp.dirichlet_bc_info]
    cons_g = np.unique(np.concatenate([ns * dm + d for ns, d in cons_nodes]))
    rng = np.random.default_rng(5)
    b_g = rng.standard_normal(n)
    x_g = rng.standard_normal(n)

    # ---- single context on the whole mesh
    ctx = gpu_ctx_factory()
    ctx.set_mesh(inp.nodes, el)
    ctx.set_element(inp.ELE)
    ctx.set_material(mat)
    ctx.build_pattern()
    ctx.upload(be.VEC_DOF, u_g)
    ctx.assemble_K(be.VEC_DOF)
    ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
    f_ref = ctx.download(be.VEC_FORCE)
    ctx.upload(be.VEC_TMP0, x_g)
    ctx.spmv(be.VEC_TMP0, be.VEC_TMP1)
    y_ref = ctx.download(be.VEC_TMP1)
    ctx.upload(be.VEC_RESIDUAL, b_g)
    ctx.dirichlet_newton(cons_g, be.VEC_RESIDUAL)
    it_ref, r0_ref, rmax_ref = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-8)
    sol_ref = ctx.download(be.VEC_X)
    hist_ref = []
    for maxit in (1, 7, 25):
        hist_ref.append((ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=maxit), ctx.download(be.VEC_X)))

    # ---- nranks contexts, one thread each
    uid = be.Context.comm_local_id()

    def rank_main(r):
        p = parts[r]
        c = setup_rank(be, p, inp, mat, uid)
        try:
            c.upload(be.VEC_DOF, p.scatter_global(u_g))
            c.assemble_K(be.VEC_DOF)
            c.internal_force(be.VEC_DOF, be.VEC_FORCE)       # sub-assembled forces, summed over the interface
            f = c.download(be.VEC_FORCE)
            c.upload(be.VEC_TMP0, p.scatter_global(x_g))
            c.spmv(be.VEC_TMP0, be.VEC_TMP1)                  # includes the interface exchange
            y = c.download(be.VEC_TMP1)
            c.upload(be.VEC_RESIDUAL, p.scatter_global(b_g))
            cons = np.unique(np.concatenate([p.localize_nodes(ns) * dm + d for ns, d in cons_nodes]))
            c.dirichlet_newton(cons, be.VEC_RESIDUAL)
            res = c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-8)
            sol = c.download(be.VEC_X)
            hist = []
            for maxit in (1, 7, 25):
                hist.append((c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=maxit), c.download(be.VEC_X)))
            return f, y, res, sol, hist
        finally:
            c.close()

    outs = run_ranks(nranks, rank_main)
    scale_f, scale_y = np.abs(f_ref).max(), np.abs(y_ref).max()
    for p, (f, y, res, sol, hist) in zip(parts, outs):
        # replicated quantities agree with the global ones on every rank that holds the node
        assert np.abs(f - p.scatter_global(f_ref)).max() < 1e-11 * scale_f
        assert np.abs(y - p.scatter_global(y_ref)).max() < 1e-12 * scale_y
        it, r0, rmax = res
        assert abs(it - it_ref) <= max(2, it_ref // 50) and abs(r0 - r0_ref) <= 1e-12 * r0_ref
        assert np.linalg.norm(sol - p.scatter_global(sol_ref)) <= 1e-6 * np.linalg.norm(sol_ref)
        # fixed iteration counts: the same recurrence, iterate by iterate
        for ((k, r0k, rmaxk), xk), ((kr, _, rmaxr), xr) in zip(hist, hist_ref):
            assert k == kr
            assert abs(rmaxk - rmaxr) <= 1e-8 * rmaxr
            assert np.linalg.norm(xk - p.scatter_global(xr)) <= 1e-9 * np.linalg.norm(xr)
    # all ranks got identical scalars (rank-ordered sums) and the gathered solution is the global one
    assert len({o[2] for o in outs}) == 1
    sol_g = partition.gather_owned(parts, [o[3] for o in outs], n)
    assert np.linalg.norm(sol_g - sol_ref) <= 1e-6 * np.linalg.norm(sol_ref)
    # ... and it solves the oracle's system
    topo = orc.Topology(inp.nodes, el, elem_def(et))
    K = orc.assemble_K(topo, u_g, oracle_material(mat).C).tolil()
    rhs = b_g.copy()
    rhs[cons_g] = 0.0
    K[cons_g, :] = 0.0
    K[:, cons_g] = 0.0
    K[cons_g, cons_g] = 1.0
    resid = K.tocsr() @ sol_g - rhs
    assert np.abs(resid).max() <= 1e-7 * np.abs(rhs).max()


def test_rendezvous_failures_are_reported(gpu_ctx_factory):
    """a rank that joins a full group, or a second group size, is an error, not a hang."""
    from femcy_amd import backend as be, partition
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck("twist_plate_C3D4.inp"))
    el = inp.eSets["C3D4"]
    mat = list(inp.materials.values())[0]
    parts = partition.build_all_parts(inp.nodes, el, 2)
    uid = be.Context.comm_local_id()
    a, b = run_ranks(2, lambda r: setup_rank(be, parts[r], inp, mat, uid))     # comm_init is collective
    assert a.download(be.VEC_DOF).size + b.download(be.VEC_DOF).size > inp.nodes.size   # shared nodes are replicated
    c = be.Context(0)
    c.set_mesh(parts[0].nodes, parts[0].elements)
    with pytest.raises(be.FemcyError):
        c.comm_init(0, 2, uid, parts[0].iface_local_dofs, parts[0].iface_global_slot, parts[0].niface_global,
                    parts[0].owner)                                  # the group is full
    with pytest.raises(be.FemcyError):
        c.comm_init(0, 3, uid, parts[0].iface_local_dofs, parts[0].iface_global_slot, parts[0].niface_global,
                    parts[0].owner)                                  # wrong group size
    for x in (a, b, c):
        x.close()


@pytest.mark.parametrize("name,nranks,axis,exchange", [
    ("twist_plate_C3D4.inp", 2, 2, "allreduce"),                     # nlgeom, *Boundary user
    ("cook_3d_linearEl_largeDef.inp", 3, 0, "neighbour"),            # neo-Hookean + *Dsload, send/recv exchange
    ("ellip_CPS8.inp", 2, 0, "auto"),                                # linear: 0/1 elimination; tuned exchange
    ("beamDeflec_quadPSE_largeD_load800.inp", 2, 0, "neighbour")])   # CPS6 + load, cut-backs
def test_partitioned_deck_solve_equals_single_context(name, nranks, axis, exchange):
    """the whole driver (increments, modified Newton, line searches, cut-backs) with the mesh split over `nranks`
    contexts: every rank runs the reference's control flow on collective scalars, so all ranks take the same
    decisions as the un-partitioned run -- same increments, same Newton counts, same displacements."""
    from femcy_amd import backend as be, partition
    from femcy_amd.body import Body
    from femcy_amd.reader import InpInfo
    from femcy_amd.stiffnessMtrx import System_of_equations
    inp = InpInfo(deck(name))
    el = list(inp.eSets.values())[0]
    mat = list(inp.materials.values())[0]
    n = inp.nodes.size

    # (direct="pcg": the partitioned runs solve the small systems of these decks with the tight PCG -- the band
    # factorisation is single-rank -- so the reference takes the same branch: this test is about the partitioning)
    ref = System_of_equations(Body(inp.nodes, el, inp.ELE), mat, inp.geometric_nonlinear, verbose=False, direct="pcg")
    ref.solve(inp)
    u_ref = ref.dof.to_numpy()
    e_ref = ref.get_elasEng()
    umax_ref = ref.ctx.vec_absmax(be.VEC_DOF)

    parts = partition.build_all_parts(inp.nodes, el, nranks, axis=axis)
    uid = be.Context.comm_local_id()

    def rank_main(r):
        p = parts[r]
        body = Body(p.nodes, p.elements, inp.ELE)
        system = System_of_equations(body, mat, inp.geometric_nonlinear, verbose=False, part=p, comm_uid=uid,
                                     exchange=exchange)
        try:
            system.solve(partition.LocalDeck(inp, p, body))
            return (system.dof.to_numpy(), system.increments, dict(system.stats), system.get_elasEng(),
                    system.ctx.vec_absmax(be.VEC_DOF))
        finally:
            system.ctx.close()

    outs = run_ranks(nranks, rank_main)
    for u, incs, stats, energy, umax in outs:
        key = lambda i: (i["converged"], i["newton_loop"])
        assert [key(i) for i in incs] == [key(i) for i in ref.increments]
        assert stats["linear_solves"] == ref.stats["linear_solves"]
        assert abs(energy - e_ref) <= 1e-7 * abs(e_ref) and abs(umax - umax_ref) <= 1e-8 * umax_ref
    u = partition.gather_owned(parts, [o[0] for o in outs], n)
    assert np.linalg.norm(u - u_ref) <= 1e-7 * np.linalg.norm(u_ref)
    for p, o in zip(parts, outs):                 # replicated interface values agree with the owner's
        assert np.linalg.norm(o[0] - p.scatter_global(u)) <= 1e-9 * np.linalg.norm(u)


def test_main_as_one_rank_rccl_job(tmp_path):
    """`python -m femcy_amd.main` launched the way torch.distributed.run launches it (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_* in the environment), forced through the partitioned path with a 1-rank RCCL communicator: process
    group, unique-id broadcast, femcy_comm_init over RCCL, collective norms, gathered result -- everything an
    N-GPU job does except talking to a second GPU.  Same displacements as the plain single-context run."""
    import os
    import subprocess
    import sys
    from helpers import ROOT
    name = deck("beam_CPS3_disp_meshSize5.inp")
    out_a, out_b = str(tmp_path / "plain.npz"), str(tmp_path / "job.npz")
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.run([sys.executable, "-m", "femcy_amd.main", name, "--quiet", "--save", out_a], check=True, env=env,
                   cwd=ROOT, timeout=600, stdout=subprocess.DEVNULL)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547",
               FEMCY_FORCE_PARTITION="1")
    r = subprocess.run([sys.executable, "-m", "femcy_amd.main", name, "--quiet", "--save", out_b], env=env, cwd=ROOT,
                       timeout=600, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "1 ranks" in r.stdout
    a, b = np.load(out_a)["dof"], np.load(out_b)["dof"]
    assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(a)


@pytest.mark.parametrize("name,nranks,axis", [("twist_plate_C3D4.inp", 4, 2), ("twist_C3D10_coarse.inp", 3, 2),
                                              ("ellip_CPS8.inp", 2, 0)])
def test_neighbour_exchange_equals_allreduce(name, nranks, axis):
    """the two interface exchanges (packed all-reduce / send-recv with the neighbouring ranks + 8-byte all-reduce of
    d.Ad) run the same PCG: same iteration counts, iterates equal to rounding, and every replica of an interface DOF
    holds the same bits under the neighbour form (rank-ordered sums).  femcy_comm_tune measures both and all ranks
    choose alike."""
    from femcy_amd import backend as be, partition
    from femcy_amd.reader import InpInfo
    inp = InpInfo(deck(name))
    el = list(inp.eSets.values())[0]
    mat = list(inp.materials.values())[0]
    dm = inp.nodes.shape[1]
    n = inp.nodes.size
    parts = partition.build_all_parts(inp.nodes, el, nranks, axis=axis)
    cons_nodes = [(np.asarray(b["node_set"]), b["dof"]) for b in inp.dirichlet_bc_info]
    b_g = np.random.default_rng(11).standard_normal(n)
    uid = be.Context.comm_local_id()

    def rank_main(r):
        p = parts[r]
        c = setup_rank(be, p, inp, mat, uid)
        try:
            c.comm_set_neighbours(p)
            c.assemble_K(-1)
            cons = np.unique(np.concatenate([p.localize_nodes(ns) * dm + d for ns, d in cons_nodes]))
            out = {}
            for mode in (0, 1):
                c.set_option(be.OPT_EXCHANGE, mode)
                c.upload(be.VEC_RESIDUAL, p.scatter_global(b_g))
                c.dirichlet_newton(cons, be.VEC_RESIDUAL)
                c.upload(be.VEC_TMP0, p.scatter_global(b_g))
                c.spmv(be.VEC_TMP0, be.VEC_TMP1)
                y = c.download(be.VEC_TMP1)
                res = c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=40)
                out[mode] = (y, res, c.download(be.VEC_X))
            # the neighbour form runs overlapped by default (interface slices first, exchange on a second stream
            # beside the interior product); the one-stream schedule must give the same iterates
            c.set_option(be.OPT_OVERLAP, 0)
            res = c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=40)
            out[2] = (None, res, c.download(be.VEC_X))
            c.set_option(be.OPT_OVERLAP, 1)
            tune = c.comm_tune(5)
            return out, tune
        finally:
            c.close()

    outs = run_ranks(nranks, rank_main)
    for p, (out, tune) in zip(parts, outs):
        (y0, r0, x0), (y1, r1, x1) = out[0], out[1]
        assert np.abs(y1 - y0).max() <= 1e-13 * np.abs(y0).max()
        assert r0[0] == r1[0] == 40 and abs(r0[2] - r1[2]) <= 1e-9 * r0[2]
        assert np.linalg.norm(x1 - x0) <= 1e-9 * np.linalg.norm(x0)
        assert out[2][1][0] == 40 and abs(out[2][1][2] - r1[2]) <= 1e-9 * r1[2]
        assert np.linalg.norm(out[2][2] - x1) <= 1e-9 * np.linalg.norm(x1)
        assert tune["allreduce_us"] > 0 and tune["neighbour_us"] > 0          # cross-check passed on every rank
    assert len({o[1]["exchange"] for o in outs}) == 1                          # one decision for the whole job
    # replicas of shared DOFs are bit-identical under the neighbour exchange
    y_glob = {}
    for p, (out, _) in zip(parts, outs):
        gd = (p.l2g[:, None] * dm + np.arange(dm)[None, :]).ravel()
        for g, v in zip(gd[p.iface_local_dofs], out[1][0][p.iface_local_dofs]):
            assert y_glob.setdefault(int(g), v) == v


# ------------------------------------------------------------------------------------------------ round 3
@pytest.mark.parametrize("nranks,wgs", [(2, 64), (4, 32), (3, 48)])
def test_persistent_pcg_across_ranks_on_one_gpu(gpu_ctx_factory, nranks, wgs):
    """the one-launch PCG on every rank, the ranks' kernels exchanging the interface rows of Ad and the scalar
    reductions through mailboxes in each other's memory (kernels_pcg_persist.hip "persistent PCG across ranks").
    Here: N contexts of one process on ONE GPU, `wgs` workgroups each so that all kernels are co-resident, pointers
    exchanged directly (the same-process branch of femcy_comm_mailbox_import); peers' stores are system-scope, polls
    are system-scope loads of fine-grained memory -- what runs between GPUs over xGMI, minus the link.  Iterates must
    equal the single-context solve of the whole mesh and the three-launch + collective path of the same ranks."""
    from femcy_amd import backend as be, meshgen, partition
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    m = meshgen.twist_plate(24, 6, 96)                                  # 82 944 C3D4, 16 975 nodes
    nodes, el = m["nodes"], m["elements"]
    mat = LinearIsotropic(*m["elastic"])
    n = nodes.size
    cons_nodes = [(np.asarray(b["node_set"]), b["dof"]) for b in m["dirichlet_bc_info"]]
    cons_g = np.unique(np.concatenate([ns * 3 + d for ns, d in cons_nodes]))
    b_g = np.sin(np.arange(n) * 0.11) * 1e3
    ctx = gpu_ctx_factory()
    ctx.set_mesh(nodes, el)
    ctx.set_element(Element_linear_tetrahedral())
    ctx.set_material(mat)
    ctx.build_pattern()
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_RESIDUAL, b_g)
    ctx.dirichlet_newton(cons_g, be.VEC_RESIDUAL)
    ref = [(ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=k), ctx.download(be.VEC_X)) for k in (1, 7, 40)]
    ref_conv = (ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-8), ctx.download(be.VEC_X))

    parts = partition.build_all_parts(nodes, el, nranks, axis=2)
    uid = be.Context.comm_local_id()
    blobs = [None] * nranks
    gate = threading.Barrier(nranks)

    def rank_main(r):
        p = parts[r]
        c = be.Context(0)
        try:
            c.set_option(107, wgs)                                      # FEMCY_TUNE_PERSIST_WGS: all ranks' kernels co-resident
            c.set_mesh(p.nodes, p.elements)
            c.set_element(Element_linear_tetrahedral())
            c.set_material(mat)
            c.build_pattern()
            c.comm_init(p.rank, p.nranks, uid, p.iface_local_dofs, p.iface_global_slot, p.niface_global, p.owner)
            c.comm_set_neighbours(p)
            blobs[r] = c.comm_mailbox_export()
            gate.wait(timeout=60)
            c.comm_mailbox_import(blobs)
            assert c.comm_persist_agree()
            c.assemble_K(-1)
            c.upload(be.VEC_RESIDUAL, p.scatter_global(b_g))
            cons = np.unique(np.concatenate([p.localize_nodes(ns) * 3 + d for ns, d in cons_nodes]))
            c.dirichlet_newton(cons, be.VEC_RESIDUAL)
            out = {}
            for multi in (1, 0):
                c.set_option(be.OPT_PCG_PERSIST_MULTI, multi)
                t0 = c.timing()
                out[multi] = [(c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=k), c.download(be.VEC_X)) for k in (1, 7, 40)]
                out[multi].append((c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-8), c.download(be.VEC_X)))
                t1 = c.timing()
                out[multi].append((t1["solves_persist"] - t0["solves_persist"], t1["solves_three"] - t0["solves_three"],
                                   t1["barrier_timeouts"] - t0["barrier_timeouts"]))
            return out
        finally:
            c.close()

    outs = run_ranks(nranks, rank_main)
    for p, out in zip(parts, outs):
        assert out[1][-1] == (4, 0, 0), out[1][-1]                     # four solves, all persistent, no time-out
        assert out[0][-1] == (0, 4, 0), out[0][-1]
        for form in (1, 0):
            for ((k, r0k, rmk), xk), ((kr, r0r, rmr), xr) in zip(out[form][:3], ref):
                assert k == kr and abs(r0k - r0r) <= 1e-12 * r0r and abs(rmk - rmr) <= 1e-9 * rmr
                assert np.linalg.norm(xk - p.scatter_global(xr)) <= 1e-9 * np.linalg.norm(xr)
            (itc, r0c, rmc), xc = out[form][3]
            (itr, _, _), xr = ref_conv
            assert abs(itc - itr) <= max(2, itr // 50) and rmc < 1e-8 * r0c
            assert np.linalg.norm(xc - p.scatter_global(xr)) <= 1e-6 * np.linalg.norm(xr)
    # replicas are bit-identical: every rank reports the same scalars, shared nodes hold the same x
    assert len({tuple(o[1][k][0] for k in range(4)) for o in outs}) == 1
    xg = {}
    for p, o in zip(parts, outs):
        x = o[1][2][1].reshape(-1, 3)
        for a, g in enumerate(p.l2g):
            key = int(g)
            if key in xg:
                assert np.array_equal(xg[key], x[a])
            else:
                xg[key] = x[a]


@pytest.mark.parametrize("nranks,wgs", [(2, 64), (4, 32)])
def test_persistent_pcg_across_ranks_2d(gpu_ctx_factory, nranks, wgs):
    """round 6: the multi-rank persistent kernel for 2 x 2 blocks (until then 2-D meshes split over ranks took three
    launches + collectives per iteration): a CPE8 beam cut along x into `nranks` parts, N contexts of one process on one GPU
    -- iterates equal to the single-context solve and to the collective loop of the same ranks, replicas bit-identical."""
    from femcy_amd import backend as be, meshgen, partition
    from femcy_amd.element_zoo import Element_quadratic_quadrilateral
    from femcy_amd.material_zoo import LinearIsotropicPlaneStrain
    m = meshgen.beam_quad8(320, 32, plane="CPE8")                       # 10 240 CPE8, 31 457 nodes
    nodes, el = m["nodes"], m["elements"]
    mat = LinearIsotropicPlaneStrain(*m["elastic"])
    n = nodes.size
    cons_nodes = [(np.asarray(b["node_set"]), b["dof"]) for b in m["dirichlet_bc_info"]]
    cons_g = np.unique(np.concatenate([ns * 2 + d for ns, d in cons_nodes]))
    b_g = np.sin(np.arange(n) * 0.11) * 1e3
    ctx = gpu_ctx_factory()
    ctx.set_mesh(nodes, el)
    ctx.set_element(Element_quadratic_quadrilateral())
    ctx.set_material(mat)
    ctx.build_pattern()
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_RESIDUAL, b_g)
    ctx.dirichlet_newton(cons_g, be.VEC_RESIDUAL)
    ref = [(ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=k), ctx.download(be.VEC_X)) for k in (1, 7, 40)]
    parts = partition.build_all_parts(nodes, el, nranks, axis=0)
    uid = be.Context.comm_local_id()
    blobs = [None] * nranks
    gate = threading.Barrier(nranks)

    def rank_main(r):
        p = parts[r]
        c = be.Context(0)
        try:
            c.set_option(107, wgs)                                      # FEMCY_TUNE_PERSIST_WGS: all ranks' kernels co-resident
            c.set_mesh(p.nodes, p.elements)
            c.set_element(Element_quadratic_quadrilateral())
            c.set_material(mat)
            c.build_pattern()
            c.comm_init(p.rank, p.nranks, uid, p.iface_local_dofs, p.iface_global_slot, p.niface_global, p.owner)
            c.comm_set_neighbours(p)
            blobs[r] = c.comm_mailbox_export()
            gate.wait(timeout=60)
            c.comm_mailbox_import(blobs)
            assert c.comm_persist_agree()
            c.assemble_K(-1)
            c.upload(be.VEC_RESIDUAL, p.scatter_global(b_g))
            cons = np.unique(np.concatenate([p.localize_nodes(ns) * 2 + d for ns, d in cons_nodes]))
            c.dirichlet_newton(cons, be.VEC_RESIDUAL)
            out = {}
            for multi in (1, 0):
                c.set_option(be.OPT_PCG_PERSIST_MULTI, multi)
                t0 = c.timing()
                out[multi] = [(c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=k), c.download(be.VEC_X)) for k in (1, 7, 40)]
                t1 = c.timing()
                out[multi].append((t1["solves_persist"] - t0["solves_persist"], t1["solves_three"] - t0["solves_three"],
                                   t1["barrier_timeouts"] - t0["barrier_timeouts"]))
            return out
        finally:
            c.close()

    outs = run_ranks(nranks, rank_main)
    for p, out in zip(parts, outs):
        assert out[1][-1] == (3, 0, 0), out[1][-1]                     # three solves, all persistent, no time-out
        assert out[0][-1] == (0, 3, 0), out[0][-1]
        for form in (1, 0):
            for ((k, r0k, rmk), xk), ((kr, r0r, rmr), xr) in zip(out[form][:3], ref):
                assert k == kr and abs(r0k - r0r) <= 1e-12 * r0r and abs(rmk - rmr) <= 1e-9 * rmr
                assert np.linalg.norm(xk - p.scatter_global(xr)) <= 1e-9 * np.linalg.norm(xr)
    assert len({tuple(o[1][k][0] for k in range(3)) for o in outs}) == 1
    xg = {}
    for p, o in zip(parts, outs):
        x = o[1][2][1].reshape(-1, 2)
        for a_, g in enumerate(p.l2g):
            key = int(g)
            if key in xg:
                assert np.array_equal(xg[key], x[a_])
            else:
                xg[key] = x[a_]


def test_persistent_pcg_across_ranks_times_out_together(gpu_ctx_factory):
    """a spin limit of 0 on every rank: the first cross-rank poll gives up, every rank's launch ends with done = 3, the
    ranks agree through the communicator and redo the solve with the three-launch + collective loop -- same iterates
    -- and stay on that loop afterwards"""
    from femcy_amd import backend as be, meshgen, partition
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    m = meshgen.twist_plate(24, 6, 96)
    nodes, el = m["nodes"], m["elements"]
    mat = LinearIsotropic(*m["elastic"])
    cons_nodes = [(np.asarray(b["node_set"]), b["dof"]) for b in m["dirichlet_bc_info"]]
    b_g = np.sin(np.arange(nodes.size) * 0.11) * 1e3
    nranks = 2
    parts = partition.build_all_parts(nodes, el, nranks, axis=2)
    uid = be.Context.comm_local_id()
    blobs = [None] * nranks
    gate = threading.Barrier(nranks)

    def rank_main(r):
        p = parts[r]
        c = be.Context(0)
        try:
            c.set_option(107, 64)
            c.set_mesh(p.nodes, p.elements)
            c.set_element(Element_linear_tetrahedral())
            c.set_material(mat)
            c.build_pattern()
            c.comm_init(p.rank, p.nranks, uid, p.iface_local_dofs, p.iface_global_slot, p.niface_global, p.owner)
            c.comm_set_neighbours(p)
            blobs[r] = c.comm_mailbox_export()
            gate.wait(timeout=60)
            c.comm_mailbox_import(blobs)
            assert c.comm_persist_agree()
            c.assemble_K(-1)
            c.upload(be.VEC_RESIDUAL, p.scatter_global(b_g))
            cons = np.unique(np.concatenate([p.localize_nodes(ns) * 3 + d for ns, d in cons_nodes]))
            c.dirichlet_newton(cons, be.VEC_RESIDUAL)
            good = (c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=9), c.download(be.VEC_X))
            t0 = c.timing()
            c.set_option(be.TUNE_BARRIER_SPIN_LIMIT, 0)
            bad = (c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=9), c.download(be.VEC_X))
            again = (c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=9), c.download(be.VEC_X))
            t1 = c.timing()
            return good, bad, again, (t1["solves_persist"] - t0["solves_persist"], t1["solves_three"] - t0["solves_three"],
                                      t1["barrier_timeouts"] - t0["barrier_timeouts"])
        finally:
            c.close()

    for good, bad, again, counts in run_ranks(nranks, rank_main):
        assert counts == (0, 2, 1), counts
        for (res, x) in (bad, again):
            assert res[0] == good[0][0] and abs(res[2] - good[0][2]) <= 1e-9 * good[0][2]
            assert np.linalg.norm(x - good[1]) <= 1e-9 * np.linalg.norm(good[1])
