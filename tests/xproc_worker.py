"""one rank of tests/test_gpu_xproc.py: a PROCESS of its own on GPU 0, joined to the others by the shared-memory
transport (femcy_comm_shm_id).  Usage: xproc_worker.py <rank> <nranks> <uid hex> <outdir> <workgroups> <scenario>"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, nranks = int(sys.argv[1]), int(sys.argv[2])
    uid = bytes.fromhex(sys.argv[3])
    outdir, wgs, scenario = sys.argv[4], int(sys.argv[5]), sys.argv[6]
    from femcy_amd import backend as be, meshgen, partition
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    cells = tuple(int(v) for v in os.environ.get("XPROC_CELLS", "24,6,96").split(","))
    m = meshgen.twist_plate(*cells)
    nodes, el = m["nodes"], m["elements"]
    mat = LinearIsotropic(*m["elastic"])
    cons_nodes = [(np.asarray(b["node_set"]), b["dof"]) for b in m["dirichlet_bc_info"]]
    b_g = np.sin(np.arange(nodes.size) * 0.11) * 1e3
    p = partition.build_part(nodes, el, nranks, rank, axis=2)
    c = be.Context(0)
    info = {"rank": rank, "pid": os.getpid()}
    try:
        if wgs > 0:
            c.set_option(107, wgs)                       # FEMCY_TUNE_PERSIST_WGS: all ranks' kernels co-resident on ONE GPU
        c.set_mesh(p.nodes, p.elements)
        c.set_element(Element_linear_tetrahedral())
        c.set_material(mat)
        c.build_pattern()
        c.comm_init(p.rank, p.nranks, uid, p.iface_local_dofs, p.iface_global_slot, p.niface_global, p.owner)
        c.comm_set_neighbours(p)
        blob = c.comm_mailbox_export()
        info["has_ipc"] = int.from_bytes(blob[44:48], "little")
        info["finegrained"] = int.from_bytes(blob[180:184], "little")
        blobs = c.comm_allgather_host(blob)
        info["peer_pids"] = [int.from_bytes(b[16:24], "little") for b in blobs]
        c.comm_mailbox_import(blobs)
        info["agreed"] = bool(c.comm_persist_agree())
        if info["agreed"]:
            info["mailbox_round_trip_us"] = c.probe_mailbox(500)         # the solver's cross-rank reduction, between processes
        c.assemble_K(-1)
        c.upload(be.VEC_RESIDUAL, p.scatter_global(b_g))
        cons = np.unique(np.concatenate([p.localize_nodes(ns) * 3 + d for ns, d in cons_nodes]))
        c.dirichlet_newton(cons, be.VEC_RESIDUAL)
        arrays = {}
        if scenario == "iterates":
            for multi in (1, 0):
                c.set_option(be.OPT_PCG_PERSIST_MULTI, multi)
                t0 = c.timing()
                res = []
                for k in (1, 7, 40):
                    res.append(list(c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=k)))
                    arrays[f"x_{multi}_{k}"] = c.download(be.VEC_X)
                res.append(list(c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-8)))
                arrays[f"x_{multi}_conv"] = c.download(be.VEC_X)
                # the default iteration cap is the global DOF count -- beyond 2^20 the tag's iteration field wraps
                # (ADVICE r3): an explicit cap above it must still take the one-launch path
                res.append(list(c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-6, maxit=(1 << 20) + 7)))
                t1 = c.timing()
                info[f"res_{multi}"] = res
                info[f"counts_{multi}"] = [t1[k] - t0[k] for k in ("solves_persist", "solves_three", "barrier_timeouts")]
        elif scenario == "exchange":
            # the three-launch + collective loop with both interface exchanges (packed all-reduce; send / recv with the
            # slab neighbours, overlapped and not), chosen by femcy_comm_tune's measurement with its cross-check
            c.set_option(be.OPT_PCG_PERSIST_MULTI, 0)
            info["tune"] = c.comm_tune(10)
            for code, overlap in ((0, 1), (1, 1), (1, 0)):
                c.set_option(be.OPT_EXCHANGE, code)
                c.set_option(be.OPT_OVERLAP, overlap)
                info[f"res_{code}_{overlap}"] = list(c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=40))
                arrays[f"x_{code}_{overlap}"] = c.download(be.VEC_X)
            t1 = c.timing()
            info["counts"] = [t1[k] for k in ("solves_persist", "solves_three", "barrier_timeouts")]
        elif scenario == "timing":
            iters = int(os.environ.get("XPROC_ITERS", "300"))
            for multi in (1, 0):
                c.set_option(be.OPT_PCG_PERSIST_MULTI, multi)
                c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=iters)
                c.sync()
                best = 1e30
                for _ in range(3):
                    t = time.perf_counter()
                    it = c.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=iters)[0]
                    c.sync()
                    best = min(best, (time.perf_counter() - t) / max(it, 1) * 1e6)
                info[f"us_per_iter_{multi}"] = best
            t1 = c.timing()
            info["counts"] = [t1[k] for k in ("solves_persist", "solves_three", "barrier_timeouts")]
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), l2g=p.l2g, **arrays)
        info["ok"] = True
    except BaseException as e:                           # noqa: BLE001
        info["ok"] = False
        info["error"] = repr(e)
    finally:
        try:
            c.close()
        except Exception:                                # noqa: BLE001
            pass
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(info, f)
    return 0 if info.get("ok") else 1


if __name__ == "__main__":
    sys.exit(main())
