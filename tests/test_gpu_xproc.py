"""-m gpu: the multi-rank path across PROCESSES on ONE GPU.  RCCL refuses two ranks on one device, so until round 4
the cross-process branches of the persistent multi-rank PCG -- hipIpcGetMemHandle / hipIpcOpenMemHandle of the
fine-grained mailbox, kernels of different processes writing into and polling each other's HBM -- had never run before
the first real multi-GPU job.  Here N processes (tests/xproc_worker.py) share GPU 0, joined by the shared-memory
transport (femcy_comm_shm_id: host staging through a POSIX segment, no RCCL), exchange their mailbox blobs over it
(femcy_comm_allgather_host), and must produce the iterates of the single-context solve with both the one-launch path
and the three-launch + collective loop."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "xproc_worker.py")


def launch(nranks, wgs, scenario, outdir, extra_env=None, timeout=420):
    from femcy_amd import backend as be
    uid = be.Context.comm_shm_id(1 << 16)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: the only mode this driver supports
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(nranks), uid.hex(), str(outdir), str(wgs), scenario],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(nranks)]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()                                        # exactly the process started above
            out, _ = p.communicate()
            out += "\n[killed after time-out]"
        logs.append(out)
    infos = []
    for r in range(nranks):
        path = os.path.join(str(outdir), f"rank{r}.json")
        assert os.path.exists(path), f"rank {r} left no record:\n{logs[r][-2000:]}"
        infos.append(json.load(open(path)))
        assert infos[r]["ok"], f"rank {r}: {infos[r].get('error')}\n{logs[r][-2000:]}"
    return infos


@pytest.mark.parametrize("nranks,wgs", [(2, 64), (4, 32)])
def test_persistent_pcg_across_processes_on_one_gpu(gpu_ctx_factory, tmp_path, nranks, wgs):
    from femcy_amd import backend as be, meshgen
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    m = meshgen.twist_plate(24, 6, 96)
    nodes, el = m["nodes"], m["elements"]
    n = nodes.size
    cons_g = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
    b_g = np.sin(np.arange(n) * 0.11) * 1e3
    ctx = gpu_ctx_factory()
    ctx.set_mesh(nodes, el)
    ctx.set_element(Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    ctx.build_pattern()
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_RESIDUAL, b_g)
    ctx.dirichlet_newton(cons_g, be.VEC_RESIDUAL)
    ref = {k: (ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=k), ctx.download(be.VEC_X)) for k in (1, 7, 40)}
    ref_conv = (ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=1e-8), ctx.download(be.VEC_X))

    infos = launch(nranks, wgs, "iterates", tmp_path)
    pids = {i["pid"] for i in infos}
    assert len(pids) == nranks                                           # really separate processes
    for i in infos:
        assert i["finegrained"] == 1 and i["has_ipc"] == 1, i            # the fine-grained mailbox has an IPC handle
        assert i["agreed"], i                                            # ... that every peer could open
        assert 0.05 < i["mailbox_round_trip_us"] < 200.0, i              # femcy_probe_mailbox between the processes' kernels
        assert i["counts_1"] == [5, 0, 0], i["counts_1"]                 # five solves, all one-launch, no time-out
        assert i["counts_0"] == [0, 5, 0], i["counts_0"]
    for r, i in enumerate(infos):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        l2g = z["l2g"]
        gd = (l2g[:, None] * 3 + np.arange(3)[None, :]).ravel()
        for multi in (1, 0):
            for j, k in enumerate((1, 7, 40)):
                (kr, r0r, rmr), xr = ref[k]
                kk, r0k, rmk = i[f"res_{multi}"][j]
                assert kk == kr and abs(r0k - r0r) <= 1e-12 * r0r and abs(rmk - rmr) <= 1e-9 * rmr
                assert np.linalg.norm(z[f"x_{multi}_{k}"] - xr[gd]) <= 1e-9 * np.linalg.norm(xr)
            (itr, _, _), xr = ref_conv
            itc, r0c, rmc = i[f"res_{multi}"][3]
            assert abs(itc - itr) <= max(2, itr // 50) and rmc < 1e-8 * r0c
            assert np.linalg.norm(z[f"x_{multi}_conv"] - xr[gd]) <= 1e-6 * np.linalg.norm(xr)
    # every rank reports the same global scalars (replicas are bit-identical)
    assert len({json.dumps(i["res_1"]) for i in infos}) == 1


def test_both_interface_exchanges_across_processes(gpu_ctx_factory, tmp_path):
    """the three-launch + collective loop across 3 processes: packed all-reduce and neighbour send / recv (overlapped
    with the interior product, and not) through the shared-memory transport give the single-context iterates;
    femcy_comm_tune cross-checks the two forms on every rank and all ranks take the same choice"""
    from femcy_amd import backend as be, meshgen
    from femcy_amd.element_zoo import Element_linear_tetrahedral
    from femcy_amd.material_zoo import LinearIsotropic
    m = meshgen.twist_plate(24, 6, 96)
    nodes, el = m["nodes"], m["elements"]
    cons_g = np.unique(np.concatenate([np.asarray(b["node_set"]) * 3 + b["dof"] for b in m["dirichlet_bc_info"]]))
    ctx = gpu_ctx_factory()
    ctx.set_mesh(nodes, el)
    ctx.set_element(Element_linear_tetrahedral())
    ctx.set_material(LinearIsotropic(*m["elastic"]))
    ctx.build_pattern()
    ctx.assemble_K(-1)
    ctx.upload(be.VEC_RESIDUAL, np.sin(np.arange(nodes.size) * 0.11) * 1e3)
    ctx.dirichlet_newton(cons_g, be.VEC_RESIDUAL)
    (kr, r0r, rmr), xr = ctx.pcg(be.VEC_RESIDUAL, be.VEC_X, eps=0.0, maxit=40), ctx.download(be.VEC_X)
    infos = launch(3, 0, "exchange", tmp_path)
    assert len({i["tune"]["exchange"] for i in infos}) == 1                  # one decision for the whole job
    for r, i in enumerate(infos):
        assert i["tune"]["allreduce_us"] > 0 and i["tune"]["neighbour_us"] > 0, i["tune"]     # cross-check passed
        assert i["counts"][0] == 0 and i["counts"][1] == 3
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        gd = (z["l2g"][:, None] * 3 + np.arange(3)[None, :]).ravel()
        for key in ("0_1", "1_1", "1_0"):
            kk, r0k, rmk = i[f"res_{key}"]
            assert kk == kr and abs(r0k - r0r) <= 1e-12 * r0r and abs(rmk - rmr) <= 1e-9 * rmr
            assert np.linalg.norm(z[f"x_{key}"] - xr[gd]) <= 1e-9 * np.linalg.norm(xr)
