#!/usr/bin/env python
"""bench.py -- BASELINE.json's headline metric on MI355X:
    "CG iters/sec + element-stiffness assemblies/sec, 1M C3D4 elems, 1/2/4/8 GPU".

One *step* = one pass of the hot path over the synthetic twist plate at state S1 (prescribed twist
of t = 0.05 on the z=0 face), exactly what one residual evaluation + linear solve of
the reference's Newton loop does on the device:
    femcy_assemble_K            get_dsdx_and_vol + assemble_stiffnessMtrx   (995 328 C3D4 per GPU)
    femcy_apply_dirichlet_newton  0/1 treatment of K, residual rows zeroed
    femcy_pcg(eps=0, maxit=ITERS) ITERS Jacobi-PCG iterations, reference recurrence + stopping test
All inputs are resident in HBM when the timed region starts.  N GPUs: the plate is cut into N z-slabs
of 995 328 elements each (weak scaling; N=1 is BASELINE's 1M mesh k=12, N=8 is its 8M mesh k=24) with
the interface exchange + two scalar reductions per CG iteration over RCCL; every rank generates only its
own slab (femcy_amd.partition.plate_slab_part), never the global mesh.

`--workload c3d10` runs the same step on BASELINE configs[4] (C3D10 plate 48x6x72: 124 416 quadratic tets on
the same 182 845 nodes) -- single GPU only; the default `c3d4` is the configuration the metric is quoted on.

`value` = CG iterations of all steps / wall time of the whole timed region (assembly included), times
global_elements/elements-per-GPU (= N) so that it is a whole-job aggregate; `cg_iters_per_s` (PCG only) and
`assemblies_per_s` (elements/s, geometry + assembly kernels) come from HIP events on the ctx stream.

Order of the run (so that an external GPU-utilisation sampler sees one contiguous busy stretch of >= 5 s at the
end): mesh -> CPU baseline leg (rank 0, N = 1) -> device set-up -> pre-warm (`--prewarm` seconds of untimed
steps) -> W warm-up steps -> K timed steps -> HBM copy probe -> one JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--iters ITERS] [--workload c3d4|c3d10] [--no-cpu-baseline]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md, HBM section); the D2D probe below reports what
                                # a plain device copy reaches on the box the bench runs on
ELEMS_PER_GPU = {"c3d4": 995328, "c3d10": 124416}
TRAFFIC_SOURCES = ("femcy_amd/csrc/kernels_pcg.hip", "femcy_amd/csrc/kernels_pcg_persist.hip", "femcy_amd/csrc/ctx.hpp",
                   "femcy_amd/csrc/pattern.cpp")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_sha():
    """fingerprint of the sources that define the SpMV kernel and its matrix layout: profiles/spmv_traffic.json is only
    valid for the kernel it was measured on"""
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, kernel):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 +
    WRITE_SIZE, separate passes, MI355X_MICROARCH.md HBM section) -- refused (None + reason) when the kernel sources
    changed since, or when the passes were taken on a different kernel."""
    tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    if not os.path.exists(tpath):
        return None, "no profiles/spmv_traffic.json"
    try:
        doc = json.load(open(tpath))
        entry = doc.get("workloads", {}).get(workload)
        if entry is None:
            return None, f"no PMC pass recorded for workload {workload}"
        if doc.get("kernel_source_sha") != kernel_source_sha():
            return None, (f"stale: PMC passes were taken on kernel sources {doc.get('kernel_source_sha')}, "
                          f"tree has {kernel_source_sha()}")
        if kernel not in entry.get("kernel", ""):
            return None, f"PMC passes of workload {workload} were taken on {entry.get('kernel', '?')[:60]}, not on {kernel}"
        return entry["hbm_bytes_per_launch"], f"profiles/spmv_traffic.json @ {doc.get('git_head', '?')}"
    except Exception as e:                                      # noqa: BLE001
        return None, f"unreadable profiles/spmv_traffic.json: {e!r}"


def hbm_copy_probe(torch):
    """device-to-device copy of 1 GiB (read + write = 2 GiB of traffic), best of 5: the bandwidth a plain copy reaches
    on this box, printed next to the 8 TB/s spec the roofline fraction is quoted against"""
    try:
        nbytes = 1 << 30
        a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        a.zero_()
        b.copy_(a)
        torch.cuda.synchronize()
        best = 0.0
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            best = max(best, 2 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del a, b
        return best
    except Exception as e:                                      # noqa: BLE001
        log(f"[bench] HBM copy probe failed: {e!r}")
        return None


def main():
    # the contract is ONE JSON line on rank 0's stdout: libraries (RCCL prints a version banner on init) must not
    # be able to add to it, so fd 1 is pointed at stderr for the whole run and the JSON goes to the saved fd
    real_stdout = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--iters", type=int, default=500, help="PCG iterations per step")
    ap.add_argument("--workload", choices=("c3d4", "c3d10"), default="c3d4",
                    help="c3d4 = BASELINE configs[2]/[3] (the metric's configuration); c3d10 = configs[4], single GPU")
    ap.add_argument("--sample", type=int, default=16, help="time every k-th SpMV launch with HIP events (1 = all)")
    ap.add_argument("--prewarm", type=float, default=4.0, help="seconds of untimed steps before the warm-up steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-comm", action="store_true", help="N=1: still run the RCCL exchange path (1-rank communicator)")
    ap.add_argument("--force-dist", action="store_true", help="N=1: still create the torch.distributed (nccl) group and use its barrier / broadcast / all-reduce (exercises the N>1 host code on one GPU)")
    ap.add_argument("--exchange", choices=("auto", "allreduce", "neighbour"), default="auto",
                    help="N>1: interface exchange per CG iteration.  auto (default) = measure both forms at start-up "
                         "(femcy_comm_tune: cross-checks their sums, takes the maximum time over the ranks) and keep "
                         "the faster, falling back to the all-reduce if the measurement fails; allreduce = the packed "
                         "global interface vector; neighbour = send/recv with the slab neighbours (overlapped with "
                         "the product of the interior rows)")
    ap.add_argument("--cells", type=str, default=None, help="override nx,ny,nz (debug / small runs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from femcy_amd import backend as be, meshgen, partition
    from femcy_amd.element_zoo import Element_linear_tetrahedral, Element_quadratic_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    from femcy_amd.user_defined import user_dirichletBC_values

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if world != N:
        if world == 1 and N > 1:
            raise SystemExit(f"--gpus {N} needs {N} ranks: launch with python -m torch.distributed.run "
                             f"--nnodes=1 --nproc-per-node {N} --master-addr 127.0.0.1 bench.py --gpus {N}")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {N}")
    quadratic = args.workload == "c3d10"
    if quadratic and (N > 1 or args.force_comm):
        raise SystemExit("--workload c3d10 is BASELINE configs[4], a single-GPU configuration")
    # test hooks (tests/test_bench_multirank_cpu.py runs the N>1 host logic on CPU with gloo and a mock Context):
    dist_backend = os.environ.get("FEMCY_BENCH_DIST_BACKEND", "nccl")
    on_gpu = os.environ.get("FEMCY_BENCH_DEVICE", "cuda") == "cuda"
    if on_gpu:
        torch.cuda.set_device(local_rank)
    use_dist = N > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if on_gpu:
            dist.init_process_group(dist_backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(dist_backend, rank=rank, world_size=world)

    def barrier():
        if use_dist:
            dist.barrier()

    # ------------------------------------------------------------------ problem (deterministic, O(local) per rank)
    t0 = time.time()
    if args.cells:
        nx, ny, nz = tuple(int(v) for v in args.cells.split(","))
    else:
        nx, ny, nz = (48, 6, 72) if quadratic else meshgen.scaling_cells(N)
    use_comm = N > 1 or args.force_comm
    elastic = (2.0e11, 0.3)
    if use_comm:
        if nz % N == 0:
            part = partition.plate_slab_part(nx, ny, nz, N, rank)      # this rank's cell layers only
        else:                                                          # odd debug grids: cut the global mesh
            g = meshgen.twist_plate(nx, ny, nz)
            part = partition.build_part(g["nodes"], g["elements"], N, rank)
        nodes, el = part.nodes, part.elements
        bcs, _ = meshgen.twist_plate_bcs(nodes)
        ne_global, n_global = 6 * nx * ny * nz, 3 * (nx + 1) * (ny + 1) * (nz + 1)
    else:
        part = None
        # FEMCY_BENCH_RENUM=1 numbers the mid-side nodes of the C3D10 plate next to the corners they connect instead of
        # behind all corners (measured in this bench: PCG iteration -4 %, row-centric assembly +50 %; default off)
        mesh = meshgen.twist_plate(nx, ny, nz, quadratic=quadratic,
                                   renumber=quadratic and os.environ.get("FEMCY_BENCH_RENUM", "0") == "1")
        nodes, el, bcs, elastic = mesh["nodes"], mesh["elements"], mesh["dirichlet_bc_info"], mesh["elastic"]
        ne_global, n_global = el.shape[0], nodes.size
    # state S1: prescribed values of the first increment (t = 0.05) written into dof, zero elsewhere
    u = np.zeros(nodes.size)
    cons = []
    for bc in bcs:
        ids = np.asarray(bc["node_set"], dtype=np.int64)
        cons.append(ids * 3 + bc["dof"])
        if bc["user"] and ids.size:
            user_dirichletBC_values(u, ids, 3, bc["dof"], nodes, 0.05)
    cons = np.unique(np.concatenate(cons)).astype(np.int32)

    # ------------------------------------------------------------------ CPU baseline leg (rank 0, N = 1 only), FIRST:
    # it is host work (10-20 s); running it before the device leg keeps the GPU-busy part of the command contiguous
    cpu = None
    if rank == 0 and N == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(nodes, el, elastic, u, cons, "C3D10" if quadratic else "C3D4")
        except Exception as e:   # the checker must never take the GPU number down with it
            log(f"[bench] cpu_baseline failed: {e!r}")

    ctx = be.Context(local_rank)
    if os.environ.get("FEMCY_BENCH_SIGMA"):                 # tuning knob: SELL sorting window
        ctx.set_option(be.OPT_SELL_SIGMA, int(os.environ["FEMCY_BENCH_SIGMA"]))
    if os.environ.get("FEMCY_BENCH_PERSIST"):               # 0 = the three-kernel PCG loop (comparison records)
        ctx.set_option(be.OPT_PCG_PERSIST, int(os.environ["FEMCY_BENCH_PERSIST"]))
    ctx.set_mesh(nodes, el)
    ctx.set_element(Element_quadratic_tetrahedral() if quadratic else Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*elastic))
    info = ctx.build_pattern()
    if use_comm:
        uid = [be.Context.comm_unique_id() if rank == 0 else None]
        if use_dist:
            dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(rank, N, uid[0], part.iface_local_dofs, part.iface_global_slot, part.niface_global, part.owner)
        # interface exchange: measure the packed all-reduce against send/recv with the slab neighbours and keep the
        # faster one (all ranks decide alike from the maximum over the ranks); --exchange pins it
        exchange = {"exchange": "allreduce", "allreduce_us": None, "neighbour_us": None}
        if hasattr(ctx, "comm_set_neighbours"):
            ctx.comm_set_neighbours(part)
            if args.exchange == "auto":
                try:
                    exchange = ctx.comm_tune(20)
                    failed = 0
                except be.FemcyError as e:                  # the send/recv form is the newer one: never let it take the
                    log(f"[bench] rank {rank}: comm_tune failed ({e}); using the all-reduce exchange")     # run down
                    failed = 1
                if use_dist:                                # every rank must run the same exchange
                    flag = torch.tensor([failed], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
                    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                    failed = int(flag.item())
                if failed:
                    ctx.set_option(be.OPT_EXCHANGE, 0)
                    exchange = {"exchange": "allreduce", "allreduce_us": None, "neighbour_us": None, "tune": "failed"}
            else:
                ctx.set_option(be.OPT_EXCHANGE, 1 if args.exchange == "neighbour" else 0)
                exchange["exchange"] = args.exchange
    else:
        exchange = None
    n, ne = ctx.n, ctx.ne
    if rank == 0:
        log(f"[bench] {args.workload} cells {nx}x{ny}x{nz}: {ne_global} elements / {n_global} DOF global, {ne} elements "
            f"/ {n} DOF per rank, nnzb {info.nnzb}, setup {time.time()-t0:.1f}s")

    ctx.upload(be.VEC_DOF, u)
    ctx.vector(be.VEC_RHS).fill(0.0)
    ctx.internal_force(be.VEC_DOF, be.VEC_FORCE)           # multi-rank: already summed over the interface
    ctx.vec_sub(be.VEC_RESIDUAL, be.VEC_FORCE, be.VEC_RHS)           # Newton residual = f_int - rhs

    cons_set = ctx.dofset(cons)            # device-resident *Boundary DOF list (what System_of_equations uses)

    def step():
        ctx.assemble_K(be.VEC_DOF)
        ctx.dofset_dirichlet_newton(cons_set, be.VEC_RESIDUAL)
        return ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=args.iters)

    def agreed_elapsed(t0):
        dt = time.perf_counter() - t0
        if use_dist:
            box = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
            dist.all_reduce(box, op=dist.ReduceOp.MAX)
            dt = float(box.item())
        return dt

    # --exchange auto, second half: femcy_comm_tune compared the bare exchanges (and cross-checked their sums); what
    # counts is the whole iteration -- the send/recv form splits the product and adds launches, the all-reduce form
    # moves the whole interface vector through every rank -- so one untimed step is run with each, the maximum over the
    # ranks is taken, and every rank keeps the faster form.
    def timed_step_all_ranks():
        barrier()
        ctx.sync()
        t0 = time.perf_counter()
        step()
        ctx.sync()
        return agreed_elapsed(t0)

    if use_comm and args.exchange == "auto" and exchange is not None and exchange.get("tune") != "failed" \
            and exchange.get("neighbour_us") is not None and exchange["neighbour_us"] >= 0:
        trial = {}
        for name, code in (("allreduce", 0), ("neighbour", 1)):
            ctx.set_option(be.OPT_EXCHANGE, code)
            step()                                           # connections, split lists, clocks
            trial[name] = min(timed_step_all_ranks(), timed_step_all_ranks())
        pick = "neighbour" if trial["neighbour"] < trial["allreduce"] else "allreduce"
        ctx.set_option(be.OPT_EXCHANGE, 1 if pick == "neighbour" else 0)
        exchange["exchange"] = pick
        exchange["step_ms"] = {k: v * 1e3 for k, v in trial.items()}

    # untimed pre-warm on top of the W warmup steps: a fresh box needs ~1 s of load before the GPU sits at its
    # sustained clocks (first bench of a cold box measured 4-6 % low with 2 warmup steps = 46 ms of work), and an
    # external utilisation sampler needs seconds, not the 0.5 s of the timed region, to see the device busy.  A step
    # holds collectives, so the ranks must agree on the number of pre-warm steps: the stop test uses the MAX of the
    # elapsed time over the ranks.
    t_warm = time.perf_counter()
    while args.warmup > 0:
        step()
        ctx.sync()
        if agreed_elapsed(t_warm) >= args.prewarm:
            break
    ctx.set_option(be.OPT_TIMING, args.sample)    # HIP events on the ctx stream; every k-th SpMV launch is sampled
    for _ in range(args.warmup):
        step()
    ctx.timing_reset()

    def device_sync():
        if on_gpu:
            torch.cuda.synchronize()
        else:
            ctx.sync()

    barrier()
    device_sync()
    t_start = time.perf_counter()
    total_iters = 0
    for _ in range(args.steps):
        it, r0, rmax = step()
        total_iters += it
    device_sync()
    barrier()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tm = ctx.timing()
    ctx.set_option(be.OPT_TIMING, 0)
    probe = hbm_copy_probe(torch) if (on_gpu and rank == 0) else None

    # ------------------------------------------------------------------ roofline of the dominant kernel
    # algorithmic bytes (BASELINE.md / SURVEY.md 8d), padding never counted:
    #   one SpMV          8*nnz + 4*nnz/dm^2 + 4*(nn+1) + 16*n
    #   one PCG iteration the SpMV + 88*n (the fused vector updates of the reference recurrence)
    spmv_bytes = 8 * info.nnz + 4 * info.nnzb + 4 * (ctx.nn + 1) + 16 * n
    iter_bytes = spmv_bytes + 88 * n
    spmv_us = tm["spmv_ms"] * 1e3 / max(tm["spmv_launches"], 1)
    persist = tm["persist_launches"] > 0 and tm["spmv_launches"] == 0
    if persist:
        # the whole solve is ONE launch of k_pcg_persist (no SpMV launches exist): the unit of one launch is the
        # iterations it ran; its duration comes from HIP events around the launch on the ctx stream
        kernel = "k_pcg_persist"
        kernel_label = f"k_pcg_persist<{ctx.dm}> (one launch = {args.iters} PCG iterations: compute_Ad + the vector updates)"
        launch_us = tm["persist_ms"] * 1e3 / tm["persist_launches"]
        launch_bytes = iter_bytes * tm["persist_iters"] / tm["persist_launches"]
        launches = int(tm["persist_launches"])
    else:
        kernel = "k_spmv"
        kernel_label = f"k_spmv<{ctx.dm}> (compute_Ad)"
        launch_us, launch_bytes, launches = spmv_us, spmv_bytes, int(tm["spmv_launches"])
    achieved = launch_bytes / (launch_us * 1e-6) / 1e9 if launch_us > 0 else 0.0
    traffic, traffic_src = pmc_traffic(args.workload, kernel) if not args.cells else (None, "non-standard --cells")
    cg_only = total_iters / (tm["pcg_ms"] * 1e-3) if tm["pcg_ms"] > 0 else 0.0
    asm_ms = (tm["geom_ms"] + tm["assemble_ms"]) / max(tm["assemble_launches"], 1)
    per_gpu = ELEMS_PER_GPU[args.workload]
    scale = ne_global / per_gpu
    kflop_per_elem = 57.0 if quadratic else 2.9                       # as-written dense contraction
    etype = "C3D10" if quadratic else "C3D4"
    which = ("BASELINE configs[4]" if quadratic else "BASELINE configs[2] at N=1, configs[3] at N=8")

    result = {
        "metric": "CG iters/sec + element-stiffness assemblies/sec, 1M C3D4 elems, 1/2/4/8 GPU",
        "value": total_iters / elapsed * scale,
        "unit": f"CG iters/s x (global elements / {per_gpu})",
        "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"twist plate {etype} {nx}x{ny}x{nz} cells, {ne_global} elements, {n_global} DOF "
                               f"({which}), state S1 (t=0.05), "
                               f"step = assemble K + Dirichlet + {args.iters} PCG iterations",
                   "elements_per_gpu": int(ne), "cg_iters_per_step": args.iters,
                   "parallelism": f"element z-slabs x{N}, slab-local mesh generation" if N > 1 else "single GPU",
                   "interface_exchange": exchange},
        "cg_iters_per_s": cg_only * scale,
        "assemblies_per_s": ne_global / (asm_ms * 1e-3) if asm_ms > 0 else 0.0,
        "assembly_ms": asm_ms,
        "pcg_us_per_iter": tm["pcg_ms"] * 1e3 / max(total_iters, 1),
        # algorithmic flop rates: SpMV 2*nnz per launch; assembly 2.9 (C3D4) / 57 (C3D10) kflop per element
        "spmv_tflops": (2 * info.nnz / (spmv_us * 1e-6) / 1e12 if spmv_us > 0 else
                        (2 * info.nnz * total_iters / (tm["pcg_ms"] * 1e-3) / 1e12 if tm["pcg_ms"] > 0 else 0.0)),
        "assembly_tflops": kflop_per_elem * 1e3 * ne_global / (asm_ms * 1e-3) / 1e12 if asm_ms > 0 else 0.0,
        "roofline": {"kernel": kernel_label, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "peak_source": "spec (MI355X_MICROARCH.md)", "copy_probe_gbs": probe,
                     "bytes_per_launch": int(launch_bytes), "avg_launch_us": launch_us,
                     "launches_timed": launches,
                     "pcg_iteration_gbs": iter_bytes * total_iters / (tm["pcg_ms"] * 1e-3) / 1e9
                     if tm["pcg_ms"] > 0 else 0.0,
                     "pcg_iteration_frac": iter_bytes * total_iters / (tm["pcg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                     if tm["pcg_ms"] > 0 else 0.0},
    }
    if traffic is not None and launch_us > 0:
        # the rate of the bytes that actually moved between the L2s and the fabric (HBM + Infinity Cache), for comparison
        # with `achieved` (algorithmic bytes): the two differ by what the kernel kept on chip, or re-read
        result["roofline"]["traffic_gbs"] = traffic / (launch_us * 1e-6) / 1e9
        result["roofline"]["traffic_over_algorithmic"] = traffic / launch_bytes if launch_bytes else None
    if persist:
        # what the persistent kernel keeps on chip: `achieved` counts ALGORITHMIC bytes, of which the register- and
        # LDS-resident block rows and the vectors never travel after the first iteration
        result["roofline"]["note"] = ("algorithmic bytes / launch time; the kernel holds the vectors and part of the "
                                      "matrix in registers / LDS for the whole solve, so the bytes that actually move "
                                      "per iteration are fewer (see `traffic`) and the figure may exceed the HBM peak")
    if rank == 0 and N == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu
    if rank == 0:
        real_stdout.write(json.dumps(result) + "\n")
        real_stdout.flush()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(nodes, el, elastic, u, cons, etype):
    """oracle/femcy_oracle.c (the reference's algorithm as written: ELL n x W, per-entry linear search +
    atomic adds, thread-per-row SpMV, 8 vector passes + 4 reductions per CG iteration) with OpenMP on all
    host cores, on a bounded sample of the same workload: 1 assembly + ~10 s of CG iterations."""
    from oracle.c_oracle import COracle
    from oracle.elements import elem_def
    from oracle.femcy_oracle import Material
    ed = elem_def(etype)
    t0 = time.time()
    co = COracle(nodes, el, ed.dN_table(), ed.gauss_weights, Material("lin3d", elastic).C)
    setup = time.time() - t0
    co.get_dsdx_and_vol(u)
    co.assemble()                                    # warm (page faults of the ELL array)
    t = time.perf_counter()
    co.get_dsdx_and_vol(u)
    co.assemble()
    t_asm = time.perf_counter() - t
    co.zero_rows_cols_unit_diag(cons)
    f = co.internal_force(u, 0, *elastic)
    f[cons] = 0.0
    co.cg(f, eps=0.0, maxit=3)
    t = time.perf_counter()
    co.cg(f, eps=0.0, maxit=10)
    per_it = (time.perf_counter() - t) / 10
    its = int(max(20, min(2000, 10.0 / per_it)))
    t = time.perf_counter()
    _, it, _, _ = co.cg(f, eps=0.0, maxit=its)
    dt = time.perf_counter() - t
    # second, implementation-independent anchor: single-thread scipy CSR A @ x
    Kcsr = co.to_csr()
    xs = np.random.default_rng(0).standard_normal(co.n)
    Kcsr @ xs
    t = time.perf_counter()
    for _ in range(5):
        Kcsr @ xs
    t_scipy = (time.perf_counter() - t) / 5
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": it / dt, "unit": "CG iters/s", "cores": co.threads(), "kind": "port",
            "assemblies_per_s": co.ne / t_asm, "assembly_ms": t_asm * 1e3,
            "cpu_model": cpu_model, "host_threads_available": os.cpu_count(),
            "scipy_csr_spmv_per_s_1thread": 1.0 / t_scipy,
            "sample": f"same {co.ne}-element {etype} mesh and state: 1 as-written assembly ({t_asm:.2f} s) + {it} CG "
                      f"iterations ({dt:.1f} s) of oracle/femcy_oracle.c, OpenMP x{co.threads()} threads; setup "
                      f"{setup:.0f} s untimed"}


if __name__ == "__main__":
    main()
