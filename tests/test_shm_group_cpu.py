"""CPU suite: the rendezvous of the shared-memory transport (femcy_amd/csrc/shm_group.hpp -- what femcy_comm_shm_id /
femcy_comm_init join processes with) driven from host arrays by 2 and 4 real processes: sums in rank order (the same
bits on every rank), gathers, a collective with mismatching lengths, a rank that never arrives (bounded wait, the group
stays broken), and nothing left behind in /dev/shm.  The device side of the transport (copies into / out of the staging
areas, the mailbox path over it) is tests/test_gpu_xproc.py."""
import ctypes as C
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "shm_group_test.cpp")


def build_lib(tmpdir):
    so = os.path.join(str(tmpdir), "libshmtest.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", so, "-lrt", "-lpthread"])
    return so


def bind(so):
    lib = C.CDLL(so)
    lib.shmtest_make_id.argtypes = [C.c_void_p, C.c_int64]
    lib.shmtest_open.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double]
    lib.shmtest_open.restype = C.c_void_p
    lib.shmtest_exchange.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
    lib.shmtest_error.argtypes = [C.c_void_p]
    lib.shmtest_error.restype = C.c_char_p
    lib.shmtest_name.argtypes = [C.c_void_p]
    lib.shmtest_name.restype = C.c_char_p
    lib.shmtest_leave.argtypes = [C.c_void_p]
    return lib


def rank_main(so, uid, rank, nranks, scenario, q):
    try:
        lib = bind(so)
        idbuf = C.create_string_buffer(uid, 128)
        h = lib.shmtest_open(rank, nranks, idbuf, 3.0 if scenario == "missing" else 30.0)
        if not h and scenario == "missing":
            # the departing rank was through before this one arrived: it marked the group broken and removed the name
            # (ShmGroup::leave), and a late rank != 0 does not found a second group -- it fails here, fast
            q.put((rank, {"name": None, "rc": -1, "err": "open failed", "rc2": -1}))
            return
        assert h, "open failed"
        name = lib.shmtest_name(h).decode()
        out = {"name": name}
        rng = np.random.default_rng(100 + rank)
        n = 1000
        mine = rng.standard_normal(n)
        if scenario == "sums":
            for rep in range(50):                                  # many back-to-back collectives: generation barrier
                v = mine * (rep + 1)
                s = np.empty(n)
                assert lib.shmtest_exchange(h, rank, v.ctypes.data, n, s.ctypes.data, 0) == 0, lib.shmtest_error(h)
                g = np.empty(n * nranks)
                assert lib.shmtest_exchange(h, rank, v.ctypes.data, n, g.ctypes.data, 1) == 0, lib.shmtest_error(h)
                out[rep] = (s.tobytes(), g.tobytes())
        elif scenario == "mismatch":
            k = n if rank else n - 1
            s = np.empty(n)
            rc = lib.shmtest_exchange(h, rank, mine.ctypes.data, k, s.ctypes.data, 0)
            out["rc"], out["err"] = rc, lib.shmtest_error(h).decode()
        elif scenario == "missing":
            if rank == nranks - 1:                                  # joins, then never takes part
                out["rc"] = 0
            else:
                s = np.empty(n)
                rc = lib.shmtest_exchange(h, rank, mine.ctypes.data, n, s.ctypes.data, 0)
                out["rc"], out["err"] = rc, lib.shmtest_error(h).decode()
                rc2 = lib.shmtest_exchange(h, rank, mine.ctypes.data, n, s.ctypes.data, 0)      # stays broken, no wait
                out["rc2"] = rc2
        lib.shmtest_leave(h)
        q.put((rank, out))
    except BaseException as e:                                      # noqa: BLE001
        q.put((rank, {"exc": repr(e)}))


def run(so, nranks, scenario, cap=4096):
    lib = bind(so)
    idbuf = C.create_string_buffer(128)
    lib.shmtest_make_id(idbuf, cap)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=rank_main, args=(so, idbuf.raw, r, nranks, scenario, q)) for r in range(nranks)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(nranks))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r in range(nranks):
        assert "exc" not in res[r], res[r]
    return res


@pytest.fixture(scope="module")
def shm_lib(tmp_path_factory):
    return build_lib(tmp_path_factory.mktemp("shmtest"))


@pytest.mark.parametrize("nranks", [2, 4])
def test_sums_and_gathers_are_the_same_bits_on_every_rank(shm_lib, nranks):
    res = run(shm_lib, nranks, "sums")
    want = [np.random.default_rng(100 + r).standard_normal(1000) for r in range(nranks)]
    for rep in (0, 17, 49):
        s0, g0 = res[0][rep]
        acc = np.zeros(1000)
        for r in range(nranks):                                     # rank order, like the transport
            acc += want[r] * (rep + 1)
        assert np.frombuffer(s0) .tobytes() == acc.tobytes()
        assert np.array_equal(np.frombuffer(g0), np.concatenate([w * (rep + 1) for w in want]))
        for r in range(1, nranks):
            assert res[r][rep] == (s0, g0)
    name = res[0]["name"]
    assert all(res[r]["name"] == name for r in range(nranks))
    assert not os.path.exists("/dev/shm" + name), "the segment's name outlived the group"


def test_mismatching_lengths_fail_on_every_rank(shm_lib):
    res = run(shm_lib, 2, "mismatch")
    for r in range(2):
        assert res[r]["rc"] != 0 and "different lengths" in res[r]["err"]
    assert not os.path.exists("/dev/shm" + res[0]["name"])


def test_a_missing_rank_times_out_and_breaks_the_group(shm_lib):
    res = run(shm_lib, 3, "missing")
    for r in range(2):
        assert res[r]["rc"] != 0 and ("timed out" in res[r]["err"] or "failed" in res[r]["err"])
        assert res[r]["rc2"] != 0
    names = {res[r]["name"] for r in range(3) if res[r].get("name")}
    assert len(names) == 1                                          # never a second, split group under the same id
    assert not os.path.exists("/dev/shm" + names.pop())
