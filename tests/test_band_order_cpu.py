"""CPU suite: the host half of the direct solve (femcy_amd/csrc/band_order.hpp) on its own -- the reverse
Cuthill-McKee order (a permutation, deterministic, every coupled pair inside the reported band, disconnected pieces and
unreferenced nodes handled, no wider than scipy's on a deck) and the band factorisation K = L S L^T with its two
sweeps (definite, indefinite, singular) against numpy.  The device kernels that do the same on tiles are checked in
tests/test_gpu_direct.py, which also runs against libfemcy_cpu.so in tests/test_cpu_backend.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "band_order_test.cpp")


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = os.path.join(str(tmp_path_factory.mktemp("band")), "libbandtest.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fopenmp", "-fPIC", "-shared", SRC, "-o", so])
    lb = C.CDLL(so)
    lb.bandtest_rcm.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lb.bandtest_rcm.restype = C.c_int32
    lb.bandtest_factor_solve.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lb.bandtest_factor_solve.restype = C.c_int64
    return lb


def rcm(lib, nn, el):
    el = np.ascontiguousarray(el, dtype=np.int32)
    rank, node_at = np.empty(nn, np.int32), np.empty(nn, np.int32)
    hb = lib.bandtest_rcm(nn, el.shape[0], el.shape[1], el.ctypes.data, rank.ctypes.data, node_at.ctypes.data)
    return hb, rank, node_at


def test_order_is_a_permutation_that_bounds_every_coupling(lib):
    rng = np.random.default_rng(0)
    # a strip of quadrilaterals numbered at random: the natural band is the whole mesh, the best one is 2 columns wide
    nx, ny = 40, 3
    ids = rng.permutation((nx + 1) * (ny + 1)).reshape(nx + 1, ny + 1)
    el = np.array([[ids[i, j], ids[i + 1, j], ids[i + 1, j + 1], ids[i, j + 1]] for i in range(nx) for j in range(ny)])
    nn = ids.size
    hb, rank, node_at = rcm(lib, nn, el)
    assert sorted(rank) == list(range(nn)) and np.array_equal(node_at[rank], np.arange(nn))
    spread = np.abs(rank[el][:, :, None] - rank[el][:, None, :]).max()
    assert spread == hb and hb <= 2 * (ny + 1) + 1                       # two columns of ny + 1 nodes (+ 1)
    hb2, rank2, _ = rcm(lib, nn, el)
    assert hb2 == hb and np.array_equal(rank2, rank)                      # deterministic


def test_disconnected_pieces_and_unreferenced_nodes(lib):
    tri = np.array([[0, 1, 2], [1, 2, 3], [10, 11, 12], [11, 12, 13]])   # two pieces; nodes 4..9, 14 belong to nothing
    hb, rank, node_at = rcm(lib, 15, tri)
    assert sorted(rank) == list(range(15))
    assert hb == np.abs(rank[tri][:, :, None] - rank[tri][:, None, :]).max() <= 3
    for piece in ((0, 1, 2, 3), (10, 11, 12, 13)):                        # a piece occupies consecutive positions
        r = np.sort(rank[list(piece)])
        assert np.array_equal(r, np.arange(r[0], r[0] + 4))


def test_band_is_no_wider_than_scipys_on_a_deck(lib):
    import sys
    sys.path.insert(0, ROOT)
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    from femcy_amd.reader import InpInfo
    inp = InpInfo(os.path.join(ROOT, "tests", "golden", "decks", "cookMembrane_CPE6_smallDef.inp"))
    el = np.asarray(list(inp.eSets.values())[0])
    nn, npe = inp.nodes.shape[0], el.shape[1]
    hb, rank, _ = rcm(lib, nn, el)
    G = sp.csr_matrix((np.ones(el.size * npe, np.int8), (np.repeat(el, npe, axis=1).ravel(), np.tile(el, (1, npe)).ravel())),
                      shape=(nn, nn))
    perm = reverse_cuthill_mckee(G, symmetric_mode=True)
    r2 = np.empty(nn, np.int64)
    r2[perm] = np.arange(nn)
    assert hb <= 1.25 * np.abs(r2[el][:, :, None] - r2[el][:, None, :]).max()
    assert hb < 0.1 * np.abs(el[:, :, None] - el[:, None, :]).max()       # the deck's own numbering is far wider


def band_of(K, bw):
    n = K.shape[0]
    A = np.zeros((n, bw + 1))
    for j in range(n):
        i1 = min(n, j + bw + 1)
        A[j, :i1 - j] = K[j:i1, j]
    return A


@pytest.mark.parametrize("kind", ["definite", "indefinite", "negative"])
def test_factorisation_and_sweeps_against_numpy(lib, kind):
    rng = np.random.default_rng(3)
    n, bw = 300, 17
    Lr = np.tril(rng.standard_normal((n, n)))
    Lr[np.abs(np.subtract.outer(np.arange(n), np.arange(n))) > bw] = 0.0
    Lr[np.arange(n), np.arange(n)] = 3.0 + rng.random(n)
    s = {"definite": np.ones(n), "negative": -np.ones(n), "indefinite": np.where(rng.random(n) < 0.2, -1.0, 1.0)}[kind]
    K = (Lr * s) @ Lr.T                                                   # K = L S L^T has the same band
    A, sg, neg = band_of(K, bw), np.empty(n), C.c_int64()
    b = rng.standard_normal(n)
    x = b.copy()
    bad = lib.bandtest_factor_solve(n, bw, A.ctypes.data, sg.ctypes.data, x.ctypes.data, C.byref(neg))
    assert bad == 0 and neg.value == int((s < 0).sum()) and np.array_equal(sg, s)
    assert np.abs(band_of(Lr, bw) - A).max() <= 1e-9 * np.abs(Lr).max()  # the factor itself (unique for a given S)
    assert np.linalg.norm(K @ x - b) <= 1e-9 * np.linalg.norm(b)


def test_zero_and_nan_pivots_are_reported(lib):
    n, bw = 20, 3
    K = np.diag(np.arange(1.0, n + 1))
    K[7, 7] = 0.0
    A, sg, neg = band_of(K, bw), np.empty(n), C.c_int64()
    x = np.ones(n)
    assert lib.bandtest_factor_solve(n, bw, A.ctypes.data, sg.ctypes.data, x.ctypes.data, C.byref(neg)) == 8
    K[7, 7] = np.nan
    A = band_of(K, bw)
    assert lib.bandtest_factor_solve(n, bw, A.ctypes.data, sg.ctypes.data, x.ctypes.data, C.byref(neg)) == 8
