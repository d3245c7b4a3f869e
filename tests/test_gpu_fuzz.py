"""Randomised parity sweep of the weighted Gauss-Newton evaluation: sizes, residual distributions
(Gaussian, heavy-tailed, skewed, bimodal, discretised with many ties), window hits and misses in
random order.  Every evaluation must equal the oracle's tree variant bit for bit, whatever
pipeline served it.  Seeds are fixed: a failure reproduces."""
import numpy as np
import pytest

import icp_rust_amd as I
import oracle_ffi as O
from parity_util import oracle_in_device_order

pytestmark = pytest.mark.gpu


def opose(T):
    return O.Pose(*[float(x) for x in T.as_array()])


def residuals(rng, n, kind):
    if kind == "gauss":
        r = rng.normal(size=(n, 2)) * rng.uniform(0.01, 0.5)
    elif kind == "heavy":
        r = rng.standard_t(2.0, size=(n, 2)) * rng.uniform(0.01, 0.2)
    elif kind == "skew":
        r = rng.gamma(2.0, 0.05, size=(n, 2)) - 0.05
    elif kind == "bimodal":
        r = rng.normal(size=(n, 2)) * 0.03 + rng.choice([-0.2, 0.25], size=(n, 2))
    elif kind == "ties":
        r = np.round(rng.normal(size=(n, 2)) * 0.1, 3)  # ~10^3 distinct values: long runs of equal keys
    elif kind == "mixed_scale":
        r = rng.normal(size=(n, 2)) * np.array([1e-4, 30.0])
    else:
        raise ValueError(kind)
    return r + rng.uniform(-0.05, 0.05, size=2)


KINDS = ["gauss", "heavy", "skew", "bimodal", "ties", "mixed_scale"]


@pytest.mark.parametrize("seed", range(18))
def test_random_sequences_of_evaluations(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(4_200, 400_000)) if seed % 3 else int(rng.integers(4_200, 30_000))
    a = rng.normal(size=(n, 2)) * rng.uniform(1.0, 40.0)
    blocks, threads = I.reduce_geometry(n)
    kind = KINDS[seed % len(KINDS)]
    r = residuals(rng, n, kind)
    c0 = I.gn_path_counters()
    for step in range(5):
        if step == 3:  # a different distribution: the window prediction is now wrong
            r = residuals(rng, n, KINDS[(seed + 2) % len(KINDS)])
        else:          # a small drift, as between inner iterations
            r = r + rng.normal(size=2) * 1e-3 * (np.abs(r).mean() + 1e-6)
        p = rng.normal(size=3) * np.array([1e-3, 1e-3, 1e-5])
        T = I.Transform(p)
        Tp = opose(T)
        # b such that residual(T, a, b) = r up to rounding: b = T a - r (the oracle sees the same inputs)
        Ta = np.stack([(Tp.r00 * a[:, 0] + Tp.r01 * a[:, 1]) + Tp.tx,
                       (Tp.r10 * a[:, 0] + Tp.r11 * a[:, 1]) + Tp.ty], axis=1)
        b = Ta - r
        got = I.weighted_gauss_newton_update(T, a, b)
        rc, want, _ = O.weighted_gauss_newton_update_tree(Tp, a, b, blocks, threads)
        if rc == O.OK:
            assert got is not None
            assert np.array_equal(got, want), (seed, step, kind, got, want)
        else:
            assert got is None
    tried, missed, short, radix, _, _ = (y - x for x, y in zip(c0, I.gn_path_counters()))
    assert tried + short + radix >= 5


@pytest.mark.parametrize("seed", range(6))
def test_random_registrations_equal_the_oracle(seed):
    """Whole estimate calls on random sub-clouds of the synthetic pair: indices, inner counts and
    pose bit-equal to the oracle in tree order (speculation hits and misses included)."""
    from icp_rust_amd import synth
    rng = np.random.default_rng(77 + seed)
    n = int(rng.integers(40_000, 90_000))
    m = int(rng.integers(40_000, 90_000))
    src, dst = synth.synthetic_pair(n, m, seed=synth.SEED + 7 * seed)
    init = I.Transform(rng.normal(size=3) * np.array([0.05, 0.05, 0.002]))
    iters = int(rng.integers(3, 9))
    icp = I.Icp3d(dst)
    T, idx, inner = icp.estimate(src, init, iters, return_info=True)
    rc, oT, oidx, oinner = oracle_in_device_order(icp, 3, dst, src, opose(init), iters)
    assert rc == O.OK
    assert np.array_equal(idx, oidx)
    assert np.array_equal(inner, oinner)
    assert np.array_equal(T.as_array(), oT.as_array())
