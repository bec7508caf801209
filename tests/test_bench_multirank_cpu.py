"""CPU suite: bench.py's N > 1 host logic (rank/env handling, partition, S1 state per rank, communicator set-up,
interface sums, weak-scaling value, one JSON line from rank 0) executed with world_size 2 over gloo.

No GPU is involved: `femcy_amd.backend.Context` is replaced -- here in the test only -- by a mock that does the same
operations with the CPU oracle and the gloo mirror of the distributed PCG (tests/dist_reference.py).  This cannot say
anything about the HIP kernels or RCCL (the -m gpu suite and the 1-rank communicator test do); it guards the Python
branch of bench.py that no single-GPU run ever executes."""
import json
import os
import socket
import sys

import numpy as np

from helpers import ROOT


class MockContext:
    """oracle-backed stand-in for femcy_amd.backend.Context, just enough for bench.py"""
    uid_calls = 0

    def __init__(self, device=0):
        self.device = device
        self.vec = {}
        self.part = None
        self.opts = {}

    # ---- definition
    def set_mesh(self, nodes, el):
        from oracle import femcy_oracle as orc
        from oracle.elements import elem_def
        self.nodes, self.el = np.asarray(nodes, float), np.asarray(el)
        self.nn, self.dm = self.nodes.shape
        self.ne, self.npe = self.el.shape
        self.n = self.nn * self.dm
        self.topo = orc.Topology(self.nodes, self.el, elem_def("C3D4"))

    def set_element(self, ELE):
        self.nGP = ELE.tables()["nGP"]

    def set_material(self, mat):
        from oracle import femcy_oracle as orc
        self.mat = orc.Material("lin3d", tuple(mat.params))

    def build_pattern(self):
        class Info:
            pass
        info = Info()
        info.nnzb = int(self.topo.adj_idx.size)
        info.nnz = info.nnzb * 9
        return info

    @staticmethod
    def comm_unique_id():
        MockContext.uid_calls += 1
        return b"\x07" * 128

    def comm_init(self, rank, nranks, uid, iface_local_dofs, iface_global_slot, niface_global, owner):
        assert uid == b"\x07" * 128 and len(owner) == self.n
        if os.environ.get("FEMCY_MOCK_FAIL_RANK") == str(rank):     # test hook: one rank dies during set-up
            raise RuntimeError(f"mock failure on rank {rank}")
        if os.environ.get("FEMCY_MOCK_HANG_RANK") == str(rank):     # test hook: one rank never returns
            import time
            time.sleep(3600)
        from types import SimpleNamespace
        self.part = SimpleNamespace(iface_local_dofs=np.asarray(iface_local_dofs), iface_global_slot=np.asarray(iface_global_slot),
                                    niface_global=niface_global, owner=np.asarray(owner))
        self.comm = (rank, nranks)

    def comm_info(self):
        return self.comm[0], self.comm[1], self.n

    # ---- persistent PCG across ranks: the host side of the mailbox set-up (blobs gathered, imported, agreed).  The
    # mock has no one-launch kernel, so it agrees and then reports the three-launch path: bench.py's cross-check must
    # see that the persistent path did not run and switch it off on every rank
    mailbox_calls = []

    def comm_mailbox_export(self):
        MockContext.mailbox_calls.append("export")
        if os.environ.get("FEMCY_MOCK_FAIL_EXPORT_RANK") == os.environ.get("RANK", "0"):
            raise RuntimeError("mock: no mailbox on this rank")
        return bytes([self.device]) * 256

    def comm_mailbox_import(self, blobs):
        assert len(blobs) == int(os.environ["WORLD_SIZE"]) and all(len(b) == 256 for b in blobs)
        assert [b[0] for b in blobs] == list(range(len(blobs)))           # rank order
        MockContext.mailbox_calls.append("import")

    def comm_persist_agree(self):
        MockContext.mailbox_calls.append("agree")
        return True

    # ---- vectors
    class _Vec:
        def __init__(self, ctx, vid):
            self.ctx, self.id = ctx, vid

        def fill(self, v):
            self.ctx.vec[self.id] = np.full(self.ctx.n, float(v))

    def vector(self, vid):
        return MockContext._Vec(self, vid)

    def upload(self, vid, arr):
        self.vec[vid] = np.array(arr, float)

    def vec_sub(self, c, a, b):
        self.vec[c] = self.vec[a] - self.vec[b]

    def iface_sum(self, vid):
        from dist_reference import iface_sum
        self.vec[vid] = iface_sum(self.part, self.vec[vid])

    # ---- hot path
    def internal_force(self, u_vec, f_vec):
        from oracle import femcy_oracle as orc
        self.vec[f_vec] = orc.internal_force(self.topo, self.vec[u_vec], self.mat)[0]
        if self.part is not None:                     # femcy_internal_force sums over the interface itself
            self.iface_sum(f_vec)

    def assemble_K(self, u_vec):
        from oracle import femcy_oracle as orc
        self.K = orc.assemble_K(self.topo, self.vec[u_vec], self.mat.C)

    def dofset(self, dofs):
        self._sets = getattr(self, "_sets", [])
        self._sets.append(np.asarray(dofs, dtype=np.int64))
        return len(self._sets) - 1

    def dofset_dirichlet_newton(self, ds, vid):
        self.dirichlet_newton(self._sets[ds], vid)

    def dirichlet_newton(self, cons, vid):
        from oracle import femcy_oracle as orc
        cons = np.asarray(cons, dtype=np.int64)
        K = orc._zero_rows_cols_unit_diag(self.K, cons).tolil()
        for i in cons:
            K[i, i] = float(self.part.owner[i]) if self.part is not None else 1.0
        self.K = K.tocsr()
        self.vec[vid][cons] = 0.0

    def pcg(self, b_vec, x_vec, eps=1e-3, maxit=0):
        from dist_reference import distributed_pcg
        x, it, r0, rmax = distributed_pcg(self.part, self.K, self.vec[b_vec], eps=eps, maxit=maxit)
        self.vec[x_vec] = x
        return it, r0, rmax

    # ---- plumbing
    def set_option(self, k, v):
        self.opts[k] = v

    def timing_reset(self):
        pass

    def timing(self):
        return {"geom_ms": 1.0, "geom_launches": 1, "assemble_ms": 1.0, "assemble_launches": 1, "force_ms": 0.0,
                "force_launches": 0, "spmv_ms": 1.0, "spmv_launches": 1, "pcg_ms": 1.0, "pcg_iters": 1,
                "persist_ms": 0.0, "persist_launches": 0, "persist_iters": 0, "solves_three": 1, "solves_small": 0,
                "solves_persist": 0, "barrier_timeouts": 0}

    def sync(self):
        pass

    def close(self):
        pass


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), FEMCY_BENCH_DIST_BACKEND="gloo", FEMCY_BENCH_DEVICE="cpu")
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from femcy_amd import backend as be
    be.Context = MockContext                                   # the mock exists only inside this test process
    out = os.open(os.path.join(out_dir, f"stdout{rank}.txt"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    os.dup2(out, 1)
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "1", "--iters", "4", "--cells", "4,2,6",
                "--no-cpu-baseline", "--prewarm", "0"]
    import bench
    bench.main()
    np.save(os.path.join(out_dir, f"uid{rank}.npy"), np.array([MockContext.uid_calls]))


def test_bench_two_ranks_on_cpu(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    line0 = open(tmp_path / "stdout0.txt").read().strip().splitlines()
    assert len(line0) == 1                                       # exactly one JSON line, from rank 0 only
    assert open(tmp_path / "stdout1.txt").read().strip() == ""
    d = json.loads(line0[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 1 and d["higher_is_better"] is True
    assert d["value"] > 0 and d["config"]["elements_per_gpu"] == 6 * 4 * 2 * 6 // 2 and d["dtype"] == "f64"
    assert "cpu_baseline" not in d and d["roofline"]["bound"] == "hbm" and d["vs_baseline"] is None
    assert int(np.load(tmp_path / "uid0.npy")[0]) == 1 and int(np.load(tmp_path / "uid1.npy")[0]) == 0
    # the mailbox set-up ran on the ranks, the agreement said yes, and the cross-check -- the mock's solves are never
    # the persistent kernel -- switched the path off again, on every rank alike
    pm = d["config"]["persistent_pcg_across_ranks"]
    assert {k: pm[k] for k in ("enabled", "agreed", "took_persistent_path", "matches_three_launch_loop")} == \
        {"enabled": False, "agreed": True, "took_persistent_path": False, "matches_three_launch_loop": True}, pm
    # round 4: both multi-rank PCG paths are timed whatever the run ends up using, the communicator's rank count and
    # the mailbox probe (absent in the mock) are part of the record
    assert pm["us_per_iteration"]["persistent_across_ranks"] is None
    assert pm["us_per_iteration"]["three_launches_plus_collectives"] > 0
    assert pm["mailbox_round_trip_us"] is None
    assert d["config"]["interface_exchange"]["communicator_ranks"] == 2
    assert "strong_scaling" not in d                             # --cells: a debug grid has no 1 M-mesh counterpart


# ------------------------------------------------------------------------------------------------ round 3
def _self_launch(extra_env, timeout=600, extra_args=()):
    """`python bench.py --gpus 2` exactly as README.md types it -- no launcher, no WORLD_SIZE -- with the CPU test hooks
    (gloo, the mock Context above) handed down through the environment"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FEMCY_BENCH_DIST_BACKEND="gloo", FEMCY_BENCH_DEVICE="cpu",
               FEMCY_BENCH_MOCK="test_bench_multirank_cpu:MockContext",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, env.get("PYTHONPATH", "")]))
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                           "--iters", "4", "--cells", "4,2,6", "--no-cpu-baseline", "--prewarm", "0", *extra_args],
                          capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_bench_self_launches_its_ranks():
    out = _self_launch({})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["launcher"] == "self-launched torch.distributed.run"
    assert d["config"]["elements_per_gpu"] == 6 * 4 * 2 * 6 // 2


def test_bench_appends_a_strong_scaling_record():
    """a weak-scaling run on N > 1 ranks also cuts ONE mesh into N slabs (BASELINE: "1M C3D4 elems, 1/2/4/8 GPU") and
    reports it under `strong_scaling`; --scaling strong makes that the headline instead"""
    out = _self_launch({"FEMCY_BENCH_STRONG_CELLS": "4,2,8"})
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["scaling"] == "weak"
    st = d["strong_scaling"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["elements_per_gpu"] == 6 * 4 * 2 * 8 // 2
    assert st["value"] > 0 and st["persistent_pcg_across_ranks"]["us_per_iteration"]["three_launches_plus_collectives"] > 0
    out = _self_launch({}, extra_args=("--scaling", "strong"))
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["scaling"] == "strong" and "strong_scaling" not in d and d["config"]["elements_per_gpu"] == 6 * 4 * 2 * 6 // 2


def test_bench_self_launch_reports_a_failed_rank():
    out = _self_launch({"FEMCY_MOCK_FAIL_RANK": "1"})
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]


def test_bench_self_launch_times_out_on_a_hung_rank():
    """a rank that never returns from the communicator set-up: the per-phase watchdog ends it (exit code 3), the
    launcher tears the job down, the command returns non-zero within seconds instead of hanging the driver"""
    import time
    t = time.time()
    out = _self_launch({"FEMCY_MOCK_HANG_RANK": "1"}, extra_args=("--comm-timeout", "8"))
    assert out.returncode != 0 and time.time() - t < 240
    assert "did not finish within" in out.stderr


def test_bench_survives_a_rank_without_mailbox():
    """the mailbox export fails on ONE rank: that rank still joins the all-gather (with None), every rank votes for the
    RCCL loop, the run completes -- the other ranks do not hang in the collective"""
    out = _self_launch({"FEMCY_MOCK_FAIL_EXPORT_RANK": "1"}, extra_args=("--comm-timeout", "60"))
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["config"]["persistent_pcg_across_ranks"]["took_persistent_path"] is False
    assert "mailbox export failed" in out.stderr and "no mailbox on rank(s) [1]" in out.stderr
