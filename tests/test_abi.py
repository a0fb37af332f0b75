"""The C-ABI library loads without a GPU and exports every symbol include/blockcopy_hip.h declares; argument
validation runs before any device call (no GPU compute in this file)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "blockcopy_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bc_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import blockcopy.backend as bk

    return bk.load_library()


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 15, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/blockcopy_hip.h but not exported"


def test_binding_covers_the_header(lib):
    """The ctypes binding declares argtypes for every compute entry point of the header."""
    for n in declared_functions():
        fn = getattr(lib, n)
        assert fn.argtypes is not None, n


def test_abi_version_and_strings(lib):
    assert lib.bc_abi_version() == 2     # 2: ring caches keep activated values; round-2/3 entry points
    assert lib.bc_error_string(0) == b"ok"
    assert b"NULL" in lib.bc_error_string(-1)
    assert [lib.bc_op_name(i).decode() for i in range(11)] == ["split", "combine", "transfer", "pad", "combine_copy", "pad_ring", "grid_tables", "interp", "affine", "nms", "conv3x3"]
    assert lib.bc_op_name(99) == b"?"


def test_argument_validation_needs_no_gpu(lib):
    N = None
    # (call, expected code): BC_ERR_NULL -1, BC_ERR_SHAPE -2, BC_ERR_ELEM -3, BC_ERR_RANGE -4
    assert lib.bc_split(N, N, N, 4, 1, 3, 8, 8, 4, 4, N) == -1
    assert lib.bc_split(N, N, N, 4, 1, 3, 8, 10, 4, 4, N) == -2          # W % bs != 0
    assert lib.bc_split(N, N, N, -1, 1, 3, 8, 8, 4, 4, N) == -2
    assert lib.bc_split(N, N, N, 4, 1, 3, 8, 8, 4, 0, N) == -3           # unit size must be >= 1 byte
    assert lib.bc_pad(N, N, N, N, N, 2, 1, 3, 2, 2, 4, 1, 16, N) == -3    # element-typed halo ops: 1, 2, 4, 8 only
    assert lib.bc_split(N, N, N, 0, 1, 3, 8, 8, 4, 4, N) == 0            # nothing to do
    assert lib.bc_combine(N, N, N, 4, 1, 1 << 12, 1 << 10, 1 << 10, 4, 4, N) == -4   # >= 2^31 elements
    assert lib.bc_combine_copy(N, N, N, N, 1, 3, 8, 8, 4, 4, N) == -1
    assert lib.bc_combine_copy_indirect(N, N, N, 1, 3, 8, 8, 4, 4, 256, N) == -1
    assert lib.bc_combine_copy_indirect(N, N, N, 1, 3, 8, 10, 4, 4, 256, N) == -2
    assert lib.bc_transfer(N, N, N, N, 3, 1, 3, 2, 2, 4, 1, 4, N) == -1
    assert lib.bc_transfer(N, N, N, N, 0, 1, 3, 2, 2, 4, 1, 4, N) == 0
    assert lib.bc_pad(N, N, N, N, N, 2, 1, 3, 2, 2, 4, 0, 4, N) == -2     # pad < 1
    assert lib.bc_pad(N, N, N, N, N, 2, 1, 3, 2, 2, 4, 5, 4, N) == -2     # pad > bs
    assert lib.bc_pad_ring(N, N, N, N, N, 2, 1, 3, 2, 2, 4, 1, 4, N) == -1
    assert lib.bc_grid_tables(N, 4, N, N, N, N, N, N) == -1
    assert lib.bc_grid_tables_host(N, 0, N, N, N, N) == -2
    assert lib.bc_interp_bilinear(N, N, 4, 2, 2, 4, 4, 0, 0.5, 0.5, 7, N) == -3
    assert lib.bc_interp_bilinear(N, N, 4, 2, 2, 4, 4, 0, 0.5, 0.5, 0, N) == -1
    assert lib.bc_pad_ring_nhwc(N, N, N, N, N, 2, 1, 8, 2, 2, 4, 1, 4, 0, N, N, 0, N) == -1
    assert lib.bc_pad_ring_nhwc(N, N, N, N, N, 2, 1, 8, 2, 2, 4, 1, 3, 0, N, N, 0, N) == -3
    assert lib.bc_affine_act_nhwc(N, N, N, N, N, 0, 16, 8, 0, N) == -1
    assert lib.bc_interp_bilinear_nhwc(N, N, 4, 8, 2, 2, 4, 4, 0, 0.5, 0.5, 0, N) == -1
    assert lib.bc_nms_sorted(N, 5000, 0.5, N, N, N, N) == -2 and lib.bc_nms_sorted(N, 10, 0.5, N, N, N, N) == -1
    assert lib.bc_prof_read(99, None, None, None) == -2


def test_profiling_counters_idle(lib):
    n, ms, by = ctypes.c_longlong(-1), ctypes.c_double(-1), ctypes.c_double(-1)
    assert lib.bc_prof_reset() == 0 and lib.bc_prof_enable(0) == 0
    assert lib.bc_prof_read(4, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)) == 0
    assert (n.value, ms.value, by.value) == (0, 0.0, 0.0)


def test_plain_c_consumer_links():
    """tests/abi_c/abi_roundtrip.cpp uses nothing but include/blockcopy_hip.h + the HIP runtime and links against the
    library (run on the GPU by tests/test_gpu_ops.py::test_plain_c_consumer_runs)."""
    import build as bc_build

    exe = bc_build.build_abi_consumer()
    assert os.access(exe, os.X_OK)
    src = open(bc_build.ABI_DEMO_SRC).read()
    assert "torch" not in src.replace("PyTorch", "") and "Python.h" not in src
