"""The per-call sort of the source points by target-grid cell (csrc/qsort.hip, round 6: hand-written LSD radix sort with
digits of up to eleven bits; rounds 1-5 called rocprim).  Its order -- ascending (key, original index) -- IS the order the
sums of a registration are folded in (include/icp_mi355x.h: icp_last_fold_order), so it must be numpy's stable sort to
the element, whatever the keys look like."""
import ctypes as C

import numpy as np
import pytest

import icp_rust_amd as I

pytestmark = pytest.mark.gpu


def _sort(keys, bits):
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    n = len(keys)
    ko, po = np.empty(max(n, 1), dtype=np.uint32), np.empty(max(n, 1), dtype=np.uint32)
    rc = I.lib().icp_debug_sort_cells(C.c_void_p(keys.ctypes.data), n, bits, C.c_void_p(ko.ctypes.data), C.c_void_p(po.ctypes.data))
    assert rc == 0, rc
    return ko[:n], po[:n]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1023, 16384, 16385, 100_003, 1_000_000, 2_100_000, 2_200_001])
@pytest.mark.parametrize("bits", [1, 7, 11, 12, 21, 22, 23, 32])
def test_sort_equals_numpy_stable_sort(n, bits):
    rng = np.random.default_rng(n * 131 + bits)
    keys = rng.integers(0, 1 << bits, size=n, dtype=np.uint64).astype(np.uint32)
    ko, po = _sort(keys, bits)
    want = np.argsort(keys, kind="stable").astype(np.uint32)
    assert np.array_equal(po, want)
    assert np.array_equal(ko, keys[want])


@pytest.mark.parametrize("shape", ["all equal", "two values", "sorted", "reversed", "few cells", "one hot tile"])
def test_sort_on_hostile_keys(shape):
    """clouds crowded into a few cells, already sorted, reversed: the same launches, the same cost, the same order"""
    n, bits = 300_007, 21
    rng = np.random.default_rng(7)
    if shape == "all equal":
        keys = np.full(n, 12345, dtype=np.uint32)
    elif shape == "two values":
        keys = np.where(rng.random(n) < 0.5, 5, (1 << 21) - 1).astype(np.uint32)
    elif shape == "sorted":
        keys = np.sort(rng.integers(0, 1 << bits, size=n)).astype(np.uint32)
    elif shape == "reversed":
        keys = np.sort(rng.integers(0, 1 << bits, size=n))[::-1].astype(np.uint32)
    elif shape == "few cells":
        keys = rng.integers(0, 7, size=n).astype(np.uint32) * 300_001 % (1 << bits)
        keys = keys.astype(np.uint32)
    else:  # every key of one tile equal, the rest random
        keys = rng.integers(0, 1 << bits, size=n).astype(np.uint32)
        keys[16384:32768] = 77
    ko, po = _sort(keys, bits)
    want = np.argsort(keys, kind="stable").astype(np.uint32)
    assert np.array_equal(po, want) and np.array_equal(ko, keys[want])
