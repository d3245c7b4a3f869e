"""Deterministic synthetic point clouds for the benchmark configurations.

Counter-based splitmix64 (the k-th draw is a pure function of (seed, k)), doubles from
the top 53 bits -- so every rank of a multi-GPU run regenerates the same clouds without
communication.  Pure integer + IEEE arithmetic in numpy.
"""
import numpy as np

SEED = 0x1C920240807
TRUTH_PARAM = (0.30, -0.20, 0.015)  # ground-truth se(2) parameter (x, y, theta)
BOX_LO = np.array([-40.0, -40.0, -2.0])
BOX_HI = np.array([40.0, 40.0, 6.0])
NOISE_SIGMA = 0.01
_DRAWS = 8  # draws consumed per point

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, counters):
    """z_k for k in `counters` (uint64 array): state = seed + (k+1)*gamma, then the mix."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (counters.astype(np.uint64) + np.uint64(1)) * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniforms(seed, n, draws=_DRAWS, first=0):
    """(n, draws) doubles in [0, 1) for points first .. first+n-1."""
    k = (np.arange(first, first + n, dtype=np.uint64)[:, None] * np.uint64(draws) +
         np.arange(draws, dtype=np.uint64)[None, :])
    return (splitmix64(seed, k) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def box_cloud(seed, n, lo=BOX_LO, hi=BOX_HI, first=0, u=None):
    """70 % of the points on the six faces of the box (area weighted), 30 % inside it."""
    if u is None:
        u = uniforms(seed, n, first=first)
    ext = hi - lo
    p = lo[None, :] + u[:, 1:4] * ext[None, :]
    # face areas: +-z, +-x, +-y
    areas = np.array([ext[0] * ext[1]] * 2 + [ext[1] * ext[2]] * 2 + [ext[0] * ext[2]] * 2)
    cum = np.cumsum(areas) / areas.sum()
    f = (u[:, 0] - 0.3) / 0.7
    on_face = u[:, 0] >= 0.3
    face = np.searchsorted(cum, f, side="right").clip(0, 5)
    axis = np.array([2, 2, 0, 0, 1, 1])[face]
    side = np.array([0, 1, 0, 1, 0, 1])[face]
    val = np.where(side == 1, hi[axis], lo[axis])
    rows = np.nonzero(on_face)[0]
    p[rows, axis[rows]] = val[rows]
    return p


def _apply_se2(param, xy):
    x, y, th = param
    c, s = np.cos(th), np.sin(th)
    if th == 0.0:
        tx, ty = x, y
    else:
        tx = (s * x - (1.0 - c) * y) / th
        ty = ((1.0 - c) * x + s * y) / th
    return np.stack([c * xy[:, 0] - s * xy[:, 1] + tx, s * xy[:, 0] + c * xy[:, 1] + ty], axis=1), (c, s, tx, ty)


def synthetic_pair(n_src, n_dst, seed=SEED, param=TRUTH_PARAM, lo=BOX_LO, hi=BOX_HI,
                   noise=NOISE_SIGMA, src_first=0, src_count=None):
    """(src, dst) for configs 3/4: dst = box cloud; src = an independent box cloud moved by
    the inverse of the truth pose on xy plus N(0, noise^2) on x, y, z.  `src_first` /
    `src_count` return only a contiguous shard of the source cloud."""
    dst = box_cloud(seed, n_dst, lo, hi)
    cnt = n_src - src_first if src_count is None else src_count
    u = uniforms(seed + 1, cnt, first=src_first)
    s = box_cloud(seed + 1, cnt, lo, hi, u=u)
    # inverse of Exp(param): R^T (p - t)
    _, (c, sn, tx, ty) = _apply_se2(param, np.zeros((1, 2)))
    dx, dy = s[:, 0] - tx, s[:, 1] - ty
    s[:, 0], s[:, 1] = c * dx + sn * dy, -sn * dx + c * dy
    # Box-Muller on draws 4..7
    r1 = np.sqrt(-2.0 * np.log(1.0 - u[:, 4]))
    r2 = np.sqrt(-2.0 * np.log(1.0 - u[:, 6]))
    s[:, 0] += noise * r1 * np.cos(2.0 * np.pi * u[:, 5])
    s[:, 1] += noise * r1 * np.sin(2.0 * np.pi * u[:, 5])
    s[:, 2] += noise * r2 * np.cos(2.0 * np.pi * u[:, 7])
    return np.ascontiguousarray(s), np.ascontiguousarray(dst)


CONVERGING_SCALE = 1000.0   # millimetres, like the reference's scans/2d
CONVERGING_CLUTTER = 0.3    # share of source points without a counterpart in the target cloud
CONVERGING_NOISE = 0.03
CONVERGING_PARAM = (0.05, -0.03, 0.002)  # a frame-to-frame sized motion: within reach of 20 outer iterations at 1M points


def converging_pair(n_src, n_dst, seed=SEED, param=CONVERGING_PARAM, noise=CONVERGING_NOISE, clutter=CONVERGING_CLUTTER,
                    scale=CONVERGING_SCALE):
    """(src, dst, truth parameter) in the regime the reference's real scans run in: src RE-OBSERVES points of dst (a
    strided subsample moved by the inverse of the truth pose, plus noise), 30 % of it is clutter without a counterpart,
    and the coordinates are in MILLIMETRES like scans/2d -- the inner loop's absolute stopping rule |delta|^2 < 1e-6
    (src/lib.rs:60, 71-73) is then tight relative to the scale, so an outer iteration runs tens of Huber / MAD
    re-weighted updates at first and a few near convergence (the oracle on 100k points: 48, 48, 52, 57, 29, 22, 33, 20,
    11, 3, 1, ...; scans/2d: mean 9.8, max 77), and the registration converges within the 20 outer
    iterations.  The independent-samples pair of the headline applies exactly one update per outer iteration."""
    dst = box_cloud(seed, n_dst)
    n_in = int(round(n_src * (1.0 - clutter)))
    sel = (np.arange(n_in, dtype=np.int64) * n_dst) // max(n_in, 1)
    u = uniforms(seed + 2, n_src)
    s = np.empty((n_src, 3))
    s[:n_in] = dst[sel]
    s[n_in:] = BOX_LO[None, :] + u[n_in:, 1:4] * (BOX_HI - BOX_LO)[None, :]
    _, (c, sn, tx, ty) = _apply_se2(param, np.zeros((1, 2)))
    dx, dy = s[:, 0] - tx, s[:, 1] - ty
    s[:, 0], s[:, 1] = c * dx + sn * dy, -sn * dx + c * dy
    r1 = np.sqrt(-2.0 * np.log(1.0 - u[:, 4]))
    r2 = np.sqrt(-2.0 * np.log(1.0 - u[:, 6]))
    s[:, 0] += noise * r1 * np.cos(2.0 * np.pi * u[:, 5])
    s[:, 1] += noise * r1 * np.sin(2.0 * np.pi * u[:, 5])
    s[:, 2] += noise * r2 * np.cos(2.0 * np.pi * u[:, 7])
    return (np.ascontiguousarray(s * scale), np.ascontiguousarray(dst * scale),
            (param[0] * scale, param[1] * scale, param[2]))


# ---- config 2 stand-in: the scan3d packet layout (examples/scan3d.rs:9,21-23,45-69) -------
N_POINTS_IN_PACKET = 24 * 16
PACKETS_PER_FRAME = 75
ROOM_LO = np.array([-3.0, -3.0, -0.5])
ROOM_HI = np.array([3.0, 3.0, 2.0])


def synthetic_scan3d_packets(n_packets, seed=SEED + 100, motion=(0.004, -0.002, 0.0008)):
    """(n_packets, 384, 3) packets of an indoor-scale scene seen from a sensor that moves by
    `motion` (se(2) parameter) per packet; ~2 % of the returns are invalid (0, 0, 0), which
    the harness removes with the reference's `norm(p) > 0.2` filter."""
    n = n_packets * N_POINTS_IN_PACKET
    u = uniforms(seed, n)
    world = box_cloud(seed, n, ROOM_LO, ROOM_HI, u=u)
    pk = np.repeat(np.arange(n_packets), N_POINTS_IN_PACKET).astype(np.float64)
    # sensor pose at packet k = Exp(k * motion); points are expressed in the sensor frame
    th = pk * motion[2]
    c, s = np.cos(th), np.sin(th)
    tx, ty = pk * motion[0], pk * motion[1]
    dx, dy = world[:, 0] - tx, world[:, 1] - ty
    out = np.stack([c * dx + s * dy, -s * dx + c * dy, world[:, 2]], axis=1)
    out += NOISE_SIGMA * 0.2 * (u[:, 4:7] - 0.5)
    out[u[:, 7] < 0.02] = 0.0
    return out.reshape(n_packets, N_POINTS_IN_PACKET, 3)


def remove_invalid_values(points):
    """examples/scan3d.rs:63-69: keep p with norm(p) > 0.2 (norm as src/norm.rs:8-21)."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    sq = p * p  # (the same products, one contiguous pass)
    nrm = np.sqrt((sq[:, 0] + sq[:, 1]) + sq[:, 2])
    return np.compress(nrm > 0.2, p, axis=0)  # (half the time of boolean-mask indexing; contiguous)
