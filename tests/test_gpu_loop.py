"""The inner loop in one launch (gn_loop.hip: estimate_transform, src/lib.rs:59-84, for 4 096 ... 2^20 pairs): its
results must be the bits of the host-stepped pipelines -- checked here against the oracle's tree variant, evaluation
by evaluation (the oracle folds in the device's documented order), at the sizes where the launch's geometry changes,
on inputs it has to hand back, and on the inputs the reference panics on."""
import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from icp_rust_amd import _lib

pytestmark = pytest.mark.gpu


def pairs(n, seed, spread=0.05, scale=20.0, param=(0.4, -0.3, 0.02), outliers=True):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, 2)) * scale
    Tt = O.transform_new(np.array(param))
    b = O.transform_apply_many(Tt, a) + rng.normal(size=(n, 2)) * spread
    if outliers:
        k = rng.integers(0, n, size=n // 10)
        b[k] += rng.normal(size=(len(k), 2)) * 5
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def oracle_loop(a, b):
    """estimate_transform with every sum folded in the tree of icp_reduce_geometry: the device's bits"""
    blocks, threads = I.reduce_geometry(len(a))
    T = O.transform_identity()
    prev, applied = np.finfo(np.float64).max, 0
    if len(a) >= 2:
        for _ in range(200):
            rc, delta, err = O.weighted_gauss_newton_update_tree(T, a, b, blocks, threads)
            if rc != O.OK:
                break
            if (delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2] < 1e-6:
                break
            if err > prev:
                break
            prev = err
            T = O.transform_mul(O.transform_new(delta), T)
            applied += 1
    return T, applied


def run(a, b):
    c0, l0 = I.gn_path_counters(), I.gn_loop_counters()
    T, inner = I.estimate_transform(a, b, return_inner_iters=True)
    c1, l1 = I.gn_path_counters(), I.gn_loop_counters()
    return T, inner, tuple(y - x for x, y in zip(l0, l1)), tuple(y - x for x, y in zip(c0, c1))


# 4 096: the smallest launch (8 workgroups, one point per thread); 131 072 / 131 073: one / two points per thread;
# 2^20: eight per thread, the largest launch; 2^20 + 1: beyond it (stepped from the host)
@pytest.mark.parametrize("n", [4_096, 5_001, 131_072, 131_073, 300_000, 524_289, 1_048_576, 1_048_577])
def test_one_launch_inner_loop_equals_the_tree_oracle(n):
    a, b = pairs(n, n)
    I.estimate_transform(a, b)  # (whatever ran: the scratch handle now has window predictions for every kind)
    T, inner, loops, _ = run(a, b)
    oT, oapplied = oracle_loop(a, b)
    assert inner == oapplied and inner >= 2
    assert np.array_equal(T.as_array(), oT.as_array())
    if n <= 1_048_576:
        assert loops[0] >= 1 and loops[1] >= inner, loops  # served by the launch
    else:
        assert loops[0] == 0, loops


def test_a_long_inner_loop_in_millimetres():
    """coordinates in millimetres: the absolute stopping rule is tight relative to the scale and the loop runs tens of
    updates (the regime of the reference's scans): one launch, or two with a hand-back, same bits"""
    n = 200_000
    a, b = pairs(n, 3, spread=30.0, scale=20_000.0, param=(50.0, -30.0, 0.002), outliers=False)
    I.estimate_transform(a, b)
    T, inner, loops, _ = run(a, b)
    oT, oapplied = oracle_loop(a, b)
    assert inner == oapplied and inner >= 4, inner
    assert np.array_equal(T.as_array(), oT.as_array())
    assert 1 <= loops[0] <= 3 and loops[1] >= inner


def test_duplicates_on_the_median_are_handed_back_and_served():
    """30 % of the x residuals are one exact value on the median: more candidates than the launch's lists take -- it
    reports a miss (also with its widest windows), hands the evaluation back, and the host's pipelines (window -> seven
    launches -> radix) serve it.  Same bits as the oracle."""
    n = 200_001
    rng = np.random.default_rng(n)
    a = rng.normal(size=(n, 2)) * 10
    r = rng.normal(size=(n, 2)) * 0.2
    I.estimate_transform(a, a - r)  # prediction from a clean distribution of the same scale
    k = int(0.3 * n)
    r[:k, 0] = 0.0
    r[k:k + (n - k) // 2, 0] = -np.abs(r[k:k + (n - k) // 2, 0]) - 1e-3
    r[k + (n - k) // 2:, 0] = np.abs(r[k + (n - k) // 2:, 0]) + 1e-3
    b = np.ascontiguousarray(a - r)
    T, inner, loops, path = run(a, b)
    oT, oapplied = oracle_loop(a, b)
    assert inner == oapplied
    assert np.array_equal(T.as_array(), oT.as_array())
    assert loops[2] >= 1 and path[3] >= 1, (loops, path)  # handed back at least once; the radix path served it
    # ... and the scratch is back in its rest state: a clean pair runs through the launch again
    a2, b2 = pairs(150_000, 9)
    I.estimate_transform(a2, b2)
    T2, inner2, loops2, _ = run(a2, b2)
    oT2, oapplied2 = oracle_loop(a2, b2)
    assert inner2 == oapplied2 and np.array_equal(T2.as_array(), oT2.as_array()) and loops2[0] >= 1


def test_nan_residual_inside_the_launch_is_reported():  # the reference panics, src/stats.rs:12
    a, b = pairs(50_000, 5)
    I.estimate_transform(a, b)
    b[1234, 1] = np.nan
    with pytest.raises(I.IcpError) as e:
        I.estimate_transform(a, b)
    assert e.value.status == _lib.NAN_INPUT
    # and the next call is served normally
    a2, b2 = pairs(50_000, 6)
    T, inner, _, _ = run(a2, b2)
    oT, oapplied = oracle_loop(a2, b2)
    assert inner == oapplied and np.array_equal(T.as_array(), oT.as_array())


def test_statistics_that_jump_between_calls_are_caught_inside_the_launch_or_handed_back():
    """the residual distribution moves by many sigmas from one call to the next: every window prediction is wrong.
    Whatever the launch makes of it (widest windows, hand-back), the bits are the oracle's."""
    n = 120_000
    for seed, spread, shift in ((1, 0.05, 0.0), (2, 0.5, 3.0), (3, 0.002, -1.0), (4, 0.05, 0.0)):
        a, b = pairs(n, seed, spread=spread)
        b[:, 0] += shift
        T, inner, _, _ = run(a, b)
        oT, oapplied = oracle_loop(a, b)
        assert inner == oapplied, (seed, inner, oapplied)
        assert np.array_equal(T.as_array(), oT.as_array()), seed
