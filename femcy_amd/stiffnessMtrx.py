"""`System_of_equations`: the host driver of the MI355X solve path, call-compatible with the
reference class (/root/reference/stiffnessMtrx.py:19-822).

Same constructor, same public attributes (`dof`, `rhs`, `nodal_force`, `residual_nodal_force`,
`du`, `dof_old`, `F`, `cauchy_stress`, `dsdx`, `vol`, `ELE`, `body`, `dm`, ...) and the same
control flow for the increment / modified-Newton / line-search drivers (`solve`, `advance_inc`),
but every kernel is a call into libfemcy_hip.so and every field is a handle to HBM-resident data
(`.to_numpy()` downloads it).  Nothing numerical runs on the CPU except the consistent Neumann
loads (a per-facet loop in the reference too, stiffnessMtrx.py:369-411) and the user Dirichlet hook.

Deliberate deviations, all outside the arithmetic of the path:
  * `solve_dof`: the reference switches to scipy's direct `spsolve` below 1e5 DOF
    (stiffnessMtrx.py:272-276).  Here that branch is a direct solve too, on the device: reverse Cuthill-McKee +
    band Cholesky (`femcy_direct_solve`, csrc/kernels_direct.hip) -- no iteration, no tolerance.  A mesh split over
    several ranks (`part`) keeps the device PCG with a tight tolerance instead (`direct_eps`, default 1e-12 on
    max|r|/max|r0|; `direct="pcg"` selects it on one rank as well); at >= 1e5 DOF the reference's own CG settings
    (eps = 1e-3) apply.
  * windows / PNG output are not produced (`show_newton_steps`, `save2path` are accepted and ignored);
    `femcy_amd.vtk_out.write_vtk` writes the mesh, displacements and Mises stress for ParaView instead.
  * the reference raises UnboundLocalError when the very first residual is < 1e-9
    (`newton_loop` unbound, :767-822); here that case returns (True, 0).
"""
import copy
import time
from typing import Tuple

import numpy as np

from . import backend as be
from . import tiGadgets as tg
from . import user_defined as ud
from .body import Body
from .conjugateGradientSolver import ConjugateGradientSolver_rowMajor as CG
from .fields import HostField


class System_of_equations:
    """K(dof) . du = rhs / residual on one MI355X."""

    def __init__(self, body: Body, material, geometric_nonlinear: bool, device: int = 0, verbose: bool = True,
                 direct_eps: float = 1.0e-12, cg_eps: float = 1.0e-3, ctx: "be.Context" = None,
                 part=None, comm_uid: bytes = None, tangent: str = "reference", exchange: str = "allreduce",
                 gather_blobs=None, direct: str = "auto", cg_branch_from: float = 1.0e5):
        """part / comm_uid: this process (or thread) holds one element partition of the mesh
        (`femcy_amd.partition.Part`, `body` built from its local nodes / elements) and joins the communicator
        `comm_uid` (RCCL unique id, or an in-process group id); every rank then runs the same `solve`."""
        self.dm = body.dm
        self.geometric_nonlinear = geometric_nonlinear
        self.body = body
        self.elements, self.nodes = body.elements, body.nodes
        self.ELE = body.ELE
        self.material = material
        self.C = material.C
        self.verbose = verbose
        self.direct_eps, self.cg_eps = direct_eps, cg_eps
        # `solve_dof` (reference :272-276) sends systems of at least 1e5 DOF to its CG (eps = 1e-3, maxit = n); the constant
        # is a parameter here so that the CG leg can be driven on any deck (tests: cg_branch_from = 0)
        self.cg_branch_from = cg_branch_from
        if direct not in ("auto", "auto-timed", "cholesky", "pcg"):
            raise ValueError("direct must be 'auto' (default: chosen by the band of the system), 'auto-timed' (chosen by "
                             "timing both on the running system), 'cholesky' (band factorisation on the device) or 'pcg' "
                             "(tight PCG)")
        self.direct = direct
        # direct = "auto": both stand-ins for the reference's spsolve return the solution to ~1e-12, so which one runs is
        # a question of time only.  The band factorisation costs n * bandwidth^2 flops behind a chain of dependent
        # launches (~20 us per panel of 32 unknowns + the tile updates), the tight PCG iterations * (matrix bytes /
        # memory bandwidth); on 2-D decks and 3-D ones up to ~4e4 DOF the factorisation wins by 1.5 ... 20 x, on 3-D
        # meshes towards 1e5 DOF (bands of 2 000 ... 2 900 sub-diagonals) the PCG is up to 2 x faster
        # (profiles/r05_direct_limit.txt).  "auto" decides from the band alone (femcy_direct_plan: a property of the mesh,
        # the same answer on every run and every machine -- round 6; round 5 timed both solvers on the running system,
        # which made the 1e-9 ... 1e-12 difference between them, and with it borderline Newton counts, depend on the
        # load of the host).  "auto-timed" keeps that exploration for whoever wants the faster solver and accepts that.
        self._auto = {"first": None, "ms": {}, "tried": set(), "pcg_ok": True, "pick": None}
        self.direct_log = []              # the solver that served each solve of the direct branch, in order

        # ---- device state: mesh, element tables, material, sparsity pattern
        self.ctx = ctx if ctx is not None else be.Context(device)
        t0 = time.time()
        self.ctx.set_mesh(body.np_nodes, body.np_elements)
        self.ctx.set_element(self.ELE)
        self.ctx.set_material(material)
        self.pattern = self.ctx.build_pattern()
        # "reference": K = B^T C B on the current configuration with the constant C, as the reference assembles it
        # (modified Newton).  "consistent": material tangent at F + geometric stiffness (extension, outside parity).
        if tangent not in ("reference", "consistent"):
            raise ValueError("tangent must be 'reference' or 'consistent'")
        self.tangent = tangent
        self.ctx.set_option(be.OPT_TANGENT, 1 if tangent == "consistent" and geometric_nonlinear else 0)
        self.part = part
        if part is not None:
            self.ctx.comm_init(part.rank, part.nranks, comm_uid, part.iface_local_dofs, part.iface_global_slot,
                               part.niface_global, part.owner)
            self.verbose = verbose and part.rank == 0
            # interface exchange per CG iteration: "allreduce" (packed global interface vector), "neighbour"
            # (send/recv with the ranks sharing nodes) or "auto" (femcy_comm_tune measures both at start-up)
            self.ctx.comm_set_neighbours(part)
            if exchange == "auto":
                self.exchange = self.ctx.comm_tune(20)
            else:
                self.ctx.set_option(be.OPT_EXCHANGE, 1 if exchange == "neighbour" else 0)
                self.exchange = {"exchange": exchange}
            # persistent PCG across ranks (femcy.h): `gather_blobs(blob) -> [blob of rank 0, 1, ...]` is the host
            # program's all-gather (torch.distributed in distributed.py); the path is used only if every rank agrees
            # -- systems large enough for the one-launch kernel -- and a rank where the set-up fails votes "no"
            self.persistent_across_ranks = False
            if gather_blobs is not None:
                try:                                 # a rank whose export fails still joins the all-gather (with None)
                    blob = self.ctx.comm_mailbox_export()
                except be.FemcyError as e:
                    self._say(f"mailbox export failed ({e})")
                    blob = None
                try:
                    blobs = gather_blobs(blob)
                    if any(b is None for b in blobs):
                        raise be.FemcyError("a rank has no mailbox")
                    self.ctx.comm_mailbox_import(blobs)
                except be.FemcyError as e:
                    self._say(f"mailbox set-up failed ({e}): three launches + collectives per CG iteration")
                    self.ctx.set_option(be.OPT_PCG_PERSIST_MULTI, 0)
                self.persistent_across_ranks = self.ctx.comm_persist_agree()
        self._say("\033[32;1m pattern: {} DOF, {} blocks of {}x{}, ELL width {} ({:.3f} s) \033[0m".format(
            self.pattern.n, self.pattern.nnzb, self.dm, self.dm, self.pattern.ell_width, time.time() - t0))

        # ---- the reference's fields, as handles to HBM
        v = self.ctx.vector
        self.rhs, self.dof = v(be.VEC_RHS), v(be.VEC_DOF)
        self.nodal_force, self.residual_nodal_force = v(be.VEC_FORCE), v(be.VEC_RESIDUAL)
        self.du, self.dof_old = v(be.VEC_DU), v(be.VEC_DOF_OLD)
        g = self.ctx.gauss_field
        self.F, self.cauchy_stress = g(be.GP_F), g(be.GP_SIGMA)
        self.dsdx, self.vol = g(be.GP_DSDX), g(be.GP_VOL)
        self.strain, self.mises_stress, self.elsEngDens = g(be.GP_STRAIN), g(be.GP_MISES), g(be.GP_ENERGY)
        self.visualize_field = self.mises_stress
        self.nodal_vals = HostField(np.zeros((self.ctx.ne, self.ctx.npe)))
        self.elsEng = 0.0
        self.sparseMtrx_rowMajor = self          # what the reference hands to the CG class
        self.sparseIJ = None

        self.time0 = 0.
        self.time1 = 0.
        self.dt = 0.
        self.compiled = False
        self._dofsets = {}
        self._loadsets = {}
        self.cg_log = []          # one entry per CG solve: iterations, max|r0|, max|r|, increment end time
        self.stats = {"assemblies": 0, "force_evals": 0, "linear_solves": 0, "cg_iterations": 0, "direct_solves": 0,
                      "direct_rejected": 0}

    @property
    def n_system(self) -> int:
        return self.dof.shape[0] if self.part is None else self.ctx.n_global

    def _say(self, msg):
        if self.verbose:
            print(msg)

    # ----------------------------------------------------------------------------- kernels
    def get_dsdx_and_vol(self):
        """geometry is fused into the assembly / internal-force launches (same dof, same result);
        kept as a method because the reference's drivers call it."""

    def assemble_stiffnessMtrx(self):
        self.ctx.assemble_K(be.VEC_DOF)
        self.stats["assemblies"] += 1

    assemble_sparseMtrx = assemble_stiffnessMtrx
    assemble_stiffnessMtrx_faster = assemble_stiffnessMtrx

    def assemble_nodal_force_GN(self):
        """F (reference configuration) -> sigma(F) -> current-configuration grad N, vol -> nodal gather."""
        self.ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)
        self.stats["force_evals"] += 1

    def get_deformation_gradient(self):
        self.ctx.internal_force(be.VEC_DOF, be.VEC_TMP0)      # F and sigma are by-products of the element pass

    # ------------------------------------------------------------------------ linear solve
    def _linear_system(self):
        if not hasattr(self, "PCG"):
            b = self.rhs if not self.geometric_nonlinear else self.residual_nodal_force
            self.PCG = CG(spm=self.sparseMtrx_rowMajor, sparseIJ=self.sparseIJ, b=b, eps=self.cg_eps)
        return self.PCG

    def _take_solution(self):
        self.stats["linear_solves"] += 1
        self.du.copy_from(self.PCG.x)
        if not self.geometric_nonlinear:
            self.dof.copy_from(self.PCG.x)
        else:
            tg.c_equals_a_minus_b(self.dof, self.dof, self.PCG.x)      # Newton: dof -= x
        return self.du

    def solve_by_CG(self, eps=None, maxit=None):
        self._linear_system()
        self.PCG.eps = self.cg_eps if eps is None else eps
        self.PCG.re_init()
        it, r0, rmax = self.ctx.pcg(self.PCG.b.id, self.PCG.x.id, eps=self.PCG.eps,
                                    maxit=self.PCG.maxit if maxit is None else maxit)
        self.PCG.iterations, self.PCG.r0, self.PCG.rmax = it, r0, rmax
        self.PCG.converged = r0 == 0.0 or rmax < self.PCG.eps * r0
        self.stats["cg_iterations"] += it
        self.cg_log.append({"iters": it, "r0": r0, "rmax": rmax, "time1": self.time1, "converged": self.PCG.converged})
        return self._take_solution()

    # sub-diagonals above which the FIRST solve of direct = "auto" goes to the tight PCG.  Cube-like C3D4 meshes, medians of
    # five (profiles/r05_direct_mfma_update.txt): 512 sub-diagonals 3.7 ms against 14.6 (PCG), 1 328: 20.3 against 45.3,
    # 2 192: 51.0 against 59.1, 2 888: 86.4 against 69.0 -- the crossover lies near 2 500
    AUTO_WIDE_BAND = 2560
    AUTO_TRY_OTHER_MS = 5.0      # a first solve slower than this makes the second solve time the other method

    def _auto_method(self) -> str:
        """'cholesky' or 'pcg' for the next solve of the reference's direct branch.  direct = "auto": by the band
        (femcy_direct_plan), once, for the whole run.  direct = "auto-timed": the first solve by the band; if it took more
        than AUTO_TRY_OTHER_MS the next three solves alternate other / first / other, so that each method is timed twice
        (the first call of either carries one-off costs: band allocation, hipGraph capture), and the faster one (minimum
        of its samples) serves the rest of the run.  A PCG that broke down or did not converge leaves the race either way."""
        a = self._auto
        if a["pick"] is not None:
            return a["pick"]
        if a["first"] is None:
            try:
                plan = self.ctx.direct_plan()
                wide = plan["bandwidth"] > self.AUTO_WIDE_BAND
                a["plan"] = plan
            except (be.FemcyError, AttributeError):
                wide = False
            a["first"] = "pcg" if wide else "cholesky"
            if self.direct == "auto":                            # deterministic: the band decides, for the whole run
                a["pick"] = a["first"]
            return a["first"]
        first = a["first"]
        other = "pcg" if first == "cholesky" else "cholesky"
        if not a["pcg_ok"]:
            a["pick"] = "cholesky"
        elif first in a["ms"] and a["ms"][first] <= self.AUTO_TRY_OTHER_MS:
            a["pick"] = first                                    # a millisecond-sized solve: nothing to gain by exploring
        else:
            cnt = a.setdefault("samples", {})
            if cnt.get(other, 0) < cnt.get(first, 0) and cnt.get(other, 0) < 2:
                return other
            if cnt.get(first, 0) < 2:
                return first
            a["pick"] = min(a["ms"], key=a["ms"].get) if a["ms"] else first
        return a["pick"]

    def solve_by_scipy(self):
        """the reference's direct branch (`spsolve`, :219-251; the name is kept): band Cholesky on the device, or the
        tight PCG where that is faster (direct = "auto" / "auto-timed")."""
        if self.part is None and self.direct in ("auto", "auto-timed"):
            method = self._auto_method()
            a = self._auto
            timed = self.direct == "auto-timed"
            if timed:
                self.ctx.sync()
            t0 = time.perf_counter()
            if method == "pcg":
                a["tried"].add("pcg")
                # an exploratory solve: K may be the indefinite matrix of a diverging Newton iterate, on which CG breaks
                # down (NaN: FEMCY_ENUMERIC) or wanders -- the factorisation (L S L^T takes negative pivots, as the
                # reference's LU does) then serves this solve and every later one
                try:
                    du = self.solve_by_CG(eps=self.direct_eps, maxit=int(min(10 * self.n_system, 2 ** 31 - 1)))
                    failed, applied = not self.PCG.converged, True
                except be.FemcyError as e:
                    if e.status != be.FEMCY_ENUMERIC:
                        raise
                    failed, applied = True, False                # the breakdown left before _take_solution: dof untouched
                if not failed:
                    if timed:
                        a["ms"]["pcg"] = min(a["ms"].get("pcg", 1e30), (time.perf_counter() - t0) * 1e3)
                    a.setdefault("samples", {})["pcg"] = a.get("samples", {}).get("pcg", 0) + 1
                    self.direct_log.append("pcg")
                    return du
                a["pcg_ok"], a["pick"] = False, "cholesky"
                if applied:
                    self.stats["linear_solves"] -= 1
                    if self.geometric_nonlinear:
                        tg.a_equals_b_plus_c_mul_d(self.dof, self.dof, 1.0, self.PCG.x)       # undo dof -= x
                t0 = time.perf_counter()
            a["tried"].add("cholesky")
            du = self._solve_direct()
            if self.direct_info is not None and timed:
                a["ms"]["cholesky"] = min(a["ms"].get("cholesky", 1e30), (time.perf_counter() - t0) * 1e3)
            # (a rejected factorisation -- ENUMERIC, served by the tight PCG inside _solve_direct -- still counts as a sample:
            # the exploration must end)
            a.setdefault("samples", {})["cholesky"] = a.get("samples", {}).get("cholesky", 0) + 1
            return du
        return self._solve_direct()

    def _solve_direct(self):
        self.direct_info = None
        if self.part is None and self.direct in ("cholesky", "auto", "auto-timed"):
            self._linear_system()
            try:
                self.direct_info = self.ctx.direct_solve(self.PCG.b.id, self.PCG.x.id)
            except be.FemcyError as e:
                if e.status == be.FEMCY_ENOMEM:
                    print(f"femcy_amd: {e}; this system is solved by the tight PCG instead", flush=True)
                    self.direct = "pcg"
                elif e.status == be.FEMCY_ENUMERIC:
                    # K singular, or so indefinite that elimination without pivoting lost it (a Newton iterate on its
                    # way out: the increment is usually cut back a few solves later).  The reference's LU with
                    # pivoting still returns something there and the driver's path depends on it, so THIS solve gets
                    # the tight PCG; if that fails as well the breakdown goes to the caller
                    self.stats["direct_rejected"] += 1
                else:
                    raise
            else:
                self.PCG.iterations, self.PCG.converged = 0, True
                self.stats["direct_solves"] += 1
                self.direct_log.append("cholesky")
                return self._take_solution()
        # a mesh split over ranks (or direct="pcg"): CG in floating point can need more than n iterations on an
        # ill-conditioned K (nu -> 0.5), so the cap is 10 n here instead of the reference CG's n
        du = self.solve_by_CG(eps=self.direct_eps, maxit=int(min(10 * self.n_system, 2 ** 31 - 1)))
        if not self.PCG.converged:
            # a direct solver would have returned the solution (or failed loudly); an unconverged iterate must not be
            # used as if it were one: report it like a numerical breakdown, so that a Newton step is cut back
            # (advance_inc) and a linear deck fails instead of printing a wrong answer
            raise be.FemcyError("tight PCG in place of the direct solve: stopped at max|r| = {:.3e} > {:.1e} * {:.3e} "
                                "after {} iterations".format(self.PCG.rmax, self.direct_eps, self.PCG.r0,
                                                             self.PCG.iterations), status=be.FEMCY_ENUMERIC)
        self.direct_log.append("pcg")
        return du

    def solve_dof(self):
        if self.n_system < self.cg_branch_from:        # DOFs of the whole system (all ranks take the same branch)
            return self.solve_by_scipy()
        return self.solve_by_CG()

    # ----------------------------------------------------------------- boundary conditions
    @staticmethod
    def _node_ids(nodeSet):
        return np.asarray(nodeSet.to_numpy() if hasattr(nodeSet, "to_numpy") else nodeSet, dtype=np.int64)

    def _dofset(self, nodeSet, dm_specified: int) -> int:
        """device-resident DOF list of a (node set, component) pair; the reference likewise turns every node
        set into a ti.field once per solve (:656-659).  Cached by content."""
        ids = self._node_ids(nodeSet)
        key = (dm_specified, ids.size, int(ids[0]) if ids.size else -1, int(ids[-1]) if ids.size else -1, int(ids.sum()))
        for cached_ids, ds in self._dofsets.get(key, ()):        # keyed by content: solve() re-wraps the node sets on
            if np.array_equal(cached_ids, ids):                  # every call, the device lists must not pile up
                return ds
        ds = self.ctx.dofset(ids * self.dm + dm_specified)
        self._dofsets.setdefault(key, []).append((ids.copy(), ds))
        return ds

    def dirichletBC_linearEquations(self, nodeSet, dm_specified: int, sval: float):
        self.ctx.dofset_dirichlet_linear(self._dofset(nodeSet, dm_specified), sval, be.VEC_RHS)

    def dirichletBC_forNewtonMethod_kernel(self, nodeSet, dm_specified: int, sval: float):
        self.ctx.dofset_dirichlet_newton(self._dofset(nodeSet, dm_specified), be.VEC_RESIDUAL)

    def dirichletBC_forNewtonMethod(self, dirichletBCs):
        for bc in dirichletBCs:
            self.dirichletBC_dof(bc["node_set"], bc["dof"], bc["val"], bc["user"], self.time1)
            self.dirichletBC_forNewtonMethod_kernel(nodeSet=bc["node_set"], dm_specified=bc["dof"], sval=bc["val"])

    def dirichletBC_dof(self, nodeSet, dm_specified: int, sval: float, user: bool, time: float):
        if not user:
            self.dirichletBC_val(nodeSet, dm_specified, sval)
        else:
            ud.user_dirichletBC(self.dof, nodeSet, self.dm, dm_specified, self.nodes, time)

    def dirichletBC_val(self, nodeSet, dm_specified: int, sval: float):
        self.ctx.dofset_fill(self._dofset(nodeSet, dm_specified), be.VEC_DOF, sval)

    def _loadset(self, load_facets) -> int:
        """device load set of one *Dsload surface, built once per face set: owning element of each facet
        (body.boundary, reference :386) and the facet's type = position of its sorted local node tuple in the
        element plugin's facet tables (reference :388-392)."""
        key = frozenset(tuple(f) for f in load_facets)           # by content: repeated solve() calls reuse the device set
        if key not in self._loadsets:
            boundary = self.body.get_boundary()
            facets = [tuple(f) for f in load_facets]
            if not facets:       # a rank of a partitioned run that holds none of the loaded facets
                self._loadsets[key] = (self.ctx.loadset(self.ELE, np.zeros(0, np.int32), np.zeros(0, np.int32)), load_facets)
                return self._loadsets[key][0]
            elem = np.fromiter((boundary[f] for f in facets), dtype=np.int64, count=len(facets))
            fnodes = np.asarray(facets, dtype=np.int64).reshape(len(facets), -1)
            conn = self.body.np_elements[elem]                                            # [nf, npe]
            local = np.sort((conn[:, None, :] == fnodes[:, :, None]).argmax(axis=2), axis=1)   # sorted local ids
            keys = self.ELE.facet_tables()["keys"]
            type_of = {k: i for i, k in enumerate(keys)}
            ft = np.fromiter((type_of[tuple(r)] for r in local.tolist()), dtype=np.int32, count=len(facets))
            self._loadsets[key] = (self.ctx.loadset(self.ELE, elem, ft), load_facets)
        return self._loadsets[key][0]

    def neumannBC(self, load_facets, load_val: float, load_dir=np.array([])):
        """consistent nodal loads of a surface traction (dead load on the undeformed geometry), evaluated on the
        device.  rhs is refreshed on every call, as in the reference (:384)."""
        self.ctx.loadset_neumann(self._loadset(load_facets), load_val, load_dir, be.VEC_RHS)

    def impose_boundary_condition(self, boundary_conditions: dict):
        for nb in boundary_conditions["neumannBCs"]:
            self.neumannBC(nb["face_set"], load_val=nb["traction"], load_dir=nb.get("direction", np.array([])))
        for bc in boundary_conditions["dirichletBCs"]:
            if not self.geometric_nonlinear:
                self.dirichletBC_linearEquations(bc["node_set"], bc["dof"], bc["val"])
            else:   # Newton: prescribed values go into dof now, K / residual are treated later
                self.dirichletBC_dof(bc["node_set"], bc["dof"], bc["val"], bc["user"], self.time1)

    # --------------------------------------------------------------------- time stepping
    def solve(self, inp, show_newton_steps: bool = False, save2path: str = None):
        """multiple time increments with automatic cut-back (reference :647-711)."""
        max_inc, min_inc = inp.time_incs["max_inc"], inp.time_incs["min_inc"]
        max_time = inp.time_incs["max_time"]
        self.dt = inp.time_incs["ini_inc"]
        neumannBCs = copy.deepcopy(inp.neumann_bc_info)
        dirichletBCs = copy.deepcopy(inp.dirichlet_bc_info)
        for bc in dirichletBCs:
            bc["node_set"] = HostField(np.array([*bc["node_set"]]), dtype=np.int32)
        boundary_conditions = {"neumannBCs": neumannBCs, "dirichletBCs": dirichletBCs}
        self.increments = []
        kinc = -1
        while self.time1 < max_time:
            kinc += 1
            self.time1 = min(self.time0 + self.dt, max_time)
            self._say("\033[40;33;1m >>>>> kinc = {}, time0 = {}, dt = {} \033[0m".format(kinc, self.time0, self.dt))
            load_ratio = self.time1 / max_time
            for i, nb in enumerate(neumannBCs):
                nb["traction"] = inp.neumann_bc_info[i]["traction"] * load_ratio
            for i, bc in enumerate(dirichletBCs):
                bc["val"] = inp.dirichlet_bc_info[i]["val"] * load_ratio
            converged, newton_loop = self.advance_inc(inp, boundary_conditions, show_newton_steps, save2path)
            self.increments.append({"kinc": kinc, "time1": self.time1, "dt": self.dt, "converged": converged,
                                    "newton_loop": newton_loop})
            if not converged:
                self.time1 = self.time0
                self.dt /= 4.
                self.dof.copy_from(self.dof_old)
                kinc -= 1
                if self.dt < min_inc:
                    print("\033[31;1m allowable minimum dt is reached, "
                          "Newton's method not converges, solution is not found. \033[0m")
                    break
                continue
            if newton_loop <= 8:
                self.dt = min(self.dt * 1.5, max_inc)
            self.dof_old.copy_from(self.dof)
            self.time0 = self.time1

    def _residual(self, dirichletBCs):
        """nodal force + K at the current dof, residual = f_int - rhs, Newton Dirichlet treatment,
        RMS norm (the block the reference repeats at :756-759, :779-783 and in inside_relaxation)."""
        self.ctx.residual_and_K(be.VEC_DOF, be.VEC_FORCE)     # assemble_nodal_force_GN + assemble_stiffnessMtrx,
        self.stats["force_evals"] += 1                        # one element pass (same dof for both)
        self.stats["assemblies"] += 1
        tg.c_equals_a_minus_b(self.residual_nodal_force, self.nodal_force, self.rhs)
        self.dirichletBC_forNewtonMethod(dirichletBCs)
        return tg.field_norm(self.residual_nodal_force)

    def advance_inc(self, inp, boundary_conditions: dict, show_newton_steps: bool = False, save2path: str = None,
                    window=None) -> Tuple[bool, int]:
        """one time increment: linear solve, or modified Newton with the reference's "boost" and
        "damp" line searches (reference :714-822)."""
        t0 = time.time()
        self.get_dsdx_and_vol()
        self.assemble_stiffnessMtrx()
        if not self.compiled:
            self.compiled = True
            self.ctx.sync()
            self._say("\033[35;1m first assembly took {:.4f} s\033[0m".format(time.time() - t0))
        self.impose_boundary_condition(boundary_conditions)

        if not inp.geometric_nonlinear:
            self.solve_dof()
            return True, 0

        dirichletBCs = boundary_conditions["dirichletBCs"]
        pre_residual = self._residual(dirichletBCs)
        if not hasattr(self, "ini_residual"):
            self.ini_residual = pre_residual          # latched once for the whole run
        self._say("\033[40;33;1m initial residual_nodal_force = {} \033[0m".format(self.ini_residual))
        if self.ini_residual < 1.e-9:
            return True, 0

        newton_loop = -1
        while pre_residual / (self.ini_residual + 1.e-30) >= 0.01:
            newton_loop += 1
            if newton_loop >= 24:
                return False, newton_loop
            try:
                du = self.solve_dof()                 # dof -= K^-1 residual
            except be.FemcyError as e:
                if e.status != be.FEMCY_ENUMERIC:
                    raise
                # the reference's solvers would hand back Inf/NaN here and trip the NaN test below on the
                # next residual; the device PCG reports the breakdown instead -> same outcome: cut the step
                self._say("linear solve broke down (NaN/Inf), automatically recompute with smaller time step")
                return False, newton_loop
            residual = self._residual(dirichletBCs)
            if np.isnan(residual):
                self._say("NaN occurs, automatically recompute with smaller time step")
                return False, newton_loop
            self._say("\033[40;33;1m newton_loop = {}, residual_nodal_force = {} \033[0m".format(newton_loop, residual))

            # boost: keep going along du while the residual declines (but not by 10x)
            relax_loop, relaxation = -1, 1.
            while 0.1 * pre_residual < residual < pre_residual:
                new_residual = residual
                relax_loop += 1
                if relax_loop >= 10:
                    break
                tg.a_equals_b_plus_c_mul_d(self.dof, self.dof, -relaxation, du)
                residual = self._residual(dirichletBCs)
                if residual > new_residual:
                    tg.a_equals_b_plus_c_mul_d(self.dof, self.dof, +relaxation, du)
                    residual = self._residual(dirichletBCs)
                    relaxation *= 0.5

            # damp: the residual grew -> take back half of the step (at most twice)
            relax_loop, relaxation = -1, 0.5
            while residual > pre_residual:
                relax_loop += 1
                if relax_loop >= 2:
                    break
                tg.a_equals_b_plus_c_mul_d(self.dof, self.dof, (1. - relaxation), du)
                tg.field_multiply(du, relaxation)
                residual = self._residual(dirichletBCs)

            pre_residual = residual
        return True, newton_loop

    # -------------------------------------------------------------------- post-processing
    def compute_strain_stress(self):
        """F, strain (infinitesimal / Green), Cauchy stress (small deformation: constitutiveOfSmallDeform;
        nlgeom: the stress of the last force evaluation) and von Mises stress per Gauss point, on the device
        (reference :436-501).  Results: .F, .strain, .cauchy_stress, .mises_stress (`.to_numpy()`)."""
        self.ctx.compute_strain_stress(be.VEC_DOF, large=self.geometric_nonlinear)

    def get_elasEng(self):
        """total elastic energy = sum over Gauss points of elasticEnergyDensity(F) * vol (reference :592-606)."""
        self.elsEng = self.ctx.elastic_energy(be.VEC_DOF)
        return self.elsEng
