import os
import sys

import pytest

# Several contexts of ONE process on ONE GPU stand in for the ranks of a multi-GPU job (tests/test_gpu_multirank.py); the
# persistent multi-rank PCG needs their kernels to run AT THE SAME TIME.  HIP maps the streams of a process onto 4
# hardware queues by default, and two persistent kernels on one queue would wait for each other for ever (they end in
# the bounded-spin fall-back instead) -- one queue per stream, set before the HIP runtime starts.  One process per GPU
# (bench.py --gpus N, femcy_amd.main under torch.distributed.run) has one such kernel per process and needs nothing.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built library (the .so files are git-ignored): build once, in-tree, like
    # __graft_entry__.build() does (hipcc cross-compiles without a GPU).  A box without hipcc keeps going: the tests
    # that need the library then fail with the loader's own message.
    lib = os.path.join(ROOT, "femcy_amd", "libfemcy_hip.so")
    if not os.path.exists(lib):
        import subprocess
        try:
            subprocess.check_call(["bash", os.path.join(ROOT, "femcy_amd", "csrc", "build.sh")])
        except Exception as e:                                  # noqa: BLE001
            print(f"[conftest] could not build {lib}: {e}", file=sys.stderr)


# FEMCY_BACKEND=cpu runs the SAME parity tests against libfemcy_cpu.so (tests/test_cpu_backend.py does that in a child
# process as part of the CPU suite).  Tests of device-only machinery are skipped there.
DEVICE_ONLY = ("test_gpu_pcg_persist.py", "test_gpu_multirank.py", "test_gpu_fullsize.py", "test_gpu_bench_contract.py",
               "test_rowsum_diagonal_needs_partition_of_unity", "test_pcg_single_rank_communicator",
               "test_main_as_one_rank_rccl_job", "test_synthetic_twist_plate_end_to_end", "test_gpu_xproc.py",
               "test_internal_row_order_and_storage_order_are_transparent", "test_direct_solve_refuses_a_partitioned_system",
               "test_auto_chooses_between_the_factorisation_and_the_tight_pcg",      # (59 k-DOF cube: minutes on the host backend)
               "test_assemble_K_rows4_tile_writeout")                                # (a knob of one device kernel)


def pytest_collection_modifyitems(config, items):
    if os.environ.get("FEMCY_BACKEND", "hip").lower() != "cpu":
        return
    skip = pytest.mark.skip(reason="device-only machinery (FEMCY_BACKEND=cpu)")
    for item in items:
        if any(pat in item.nodeid for pat in DEVICE_ONLY):
            item.add_marker(skip)


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """creates femcy_amd Contexts; a -m gpu run on a box without a GPU must fail, not skip."""
    from femcy_amd.backend import Context
    made = []

    def make(device=0):
        c = Context(device)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()
