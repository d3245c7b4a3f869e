"""Pins the CPU oracle against every known-answer test the reference holds for the path.

The reference (Rust) cannot be built here, so its 24 unit tests are restated one for one:
same inputs, same literals, same assertions (exact equality where the reference asserts
equality).  File:line citations are under /root/reference.
"""
import math

import numpy as np

import oracle_ffi as O

PI = math.pi


def P(*v):
    return np.array(v, dtype=np.float64)


def norm(v):
    v = np.asarray(v, dtype=np.float64)
    m = np.asfortranarray(v.reshape(v.shape[0], -1))
    import ctypes as C

    return O.lib().orc_norm(m.ctypes.data_as(C.POINTER(C.c_double)), m.shape[0], m.shape[1])


# ---------------------------------------------------------------- src/lib.rs tests --


def test_residual():  # lib.rs:267-274
    T = O.transform_new(P(-10.0, 20.0, 0.01))
    src = P(7.0, 8.0)
    dst = O.transform_apply(T, src)
    assert np.array_equal(O.residual(T, src, dst), np.zeros(2))


def test_error():  # lib.rs:276-297
    src = np.array([[-6.0, 9.0], [-1.0, 9.0], [-4.0, -4.0]])
    dst = np.array([[-4.0, 4.0], [0.0, 3.0], [-3.0, -8.0]])
    T = O.transform_new(P(10.0, 20.0, 0.01))
    r = [O.residual(T, s, d) for s, d in zip(src, dst)]
    dot = lambda v: v[0] * v[0] + v[1] * v[1]
    expected = dot(r[0]) + dot(r[1]) + dot(r[2])
    assert O.error(T, src, dst) == expected


def test_gauss_newton_update_input_size():  # lib.rs:299-318
    T = O.transform_new(P(10.0, 30.0, -0.15))
    rc, _ = O.gauss_newton_update(T, np.zeros((0, 2)), np.zeros((0, 2)))
    assert rc == O.NONE
    src = np.array([[-8.89304516, 0.54202289]])
    dst = O.transform_apply_many(T, src)
    assert O.gauss_newton_update(T, src, dst)[0] == O.NONE
    src = np.array([[-8.89304516, 0.54202289], [-4.03198385, -2.81807802]])
    dst = O.transform_apply_many(T, src)
    assert O.gauss_newton_update(T, src, dst)[0] == O.OK


def test_gauss_newton_update():  # lib.rs:320-351
    true_param = P(10.0, 30.0, -0.15)
    initial_param = true_param + P(0.3, -0.5, 0.001)
    Tt = O.transform_new(true_param)
    Ti = O.transform_new(initial_param)
    src = np.array([[-8.76116663, 3.50338231], [-5.21184804, -1.91561705], [6.63141168, 4.8915293],
                    [-2.29215281, -4.72658399], [6.81352587, -0.81624617]])
    dst = O.transform_apply_many(Tt, src)
    rc, update = O.gauss_newton_update(Ti, src, dst)
    assert rc == O.OK
    Tu = O.transform_new(initial_param + update)
    e0 = O.error(Ti, src, dst)
    e1 = O.error(Tu, src, dst)
    assert e1 < e0 * 0.01


def test_weighted_gauss_newton_update_input_size():  # lib.rs:353-401
    T = O.transform_new(P(10.0, 30.0, -0.15))
    assert O.weighted_gauss_newton_update(T, np.zeros((0, 2)), np.zeros((0, 2)))[0] == O.NONE
    src = np.array([[-8.89304516, 0.54202289]])
    assert O.weighted_gauss_newton_update(T, src, O.transform_apply_many(T, src))[0] == O.NONE
    src = np.array([[-8.89304516, 0.54202289], [-4.03198385, -2.81807802]])
    assert O.weighted_gauss_newton_update(T, src, O.transform_apply_many(T, src))[0] == O.NONE
    src = np.array([[-8.89304516, 0.54202289], [-4.03198385, -2.81807802], [-4.03198385, -2.81807802]])
    assert O.weighted_gauss_newton_update(T, src, O.transform_apply_many(T, src))[0] == O.NONE
    src = np.array([[-8.89304516, 0.54202289], [-4.03198385, -2.81807802], [4.40356349, -9.43358563]])
    assert O.weighted_gauss_newton_update(T, src, O.transform_apply_many(T, src))[0] == O.NONE


def test_weighted_gauss_newton_update_zero_x_diff():  # lib.rs:403-427
    src = np.array([[0.0, 0.0], [0.0, 0.1], [0.0, 0.2], [0.0, 0.3], [0.0, 0.4], [0.0, 0.5]])
    Tt = O.transform_new(P(0.0, 0.01, 0.0))
    dst = O.transform_apply_many(Tt, src)
    Ti = O.transform_new(P(0.0, 0.0, 0.0))
    assert O.weighted_gauss_newton_update(Ti, src, dst)[0] == O.NONE


WGN_SRC = np.array([
    [-8.89304516, 0.54202289], [-4.03198385, -2.81807802], [-5.92679530, 9.62339266],
    [-4.04966218, -4.44595403], [-2.86369420, -9.13843999], [-6.97749644, -8.90180581],
    [-9.66454985, 6.32282424], [7.02264007, -0.88684585], [4.19700110, -1.42366424],
    [-0.68034875, -0.48699014], [1.89645382, 1.86119400], [7.09550743, 2.18289525],
    [-7.95383118, -5.16650913], [-5.40235599, 2.70675665], [-5.38909696, -5.48180288],
    [-9.00498232, -5.12191142], [-8.54899319, -3.25752055], [6.89969814, 3.53276123],
    [5.06875729, -0.28918540]])
WGN_NOISE = np.array([
    [0.01058790, 0.01302535], [0.01392508, 0.00835860], [0.01113885, -0.00693269],
    [0.01673124, -0.01735564], [-0.01219263, 0.00080933], [-0.00396817, 0.00111582],
    [-0.00444043, 0.00658505], [-0.01576271, -0.00701065], [0.00464000, -0.00406790],
    [0.00269374, -0.00787015], [-0.00494243, 0.00350137], [0.00343766, -0.00039311],
    [0.00661565, -0.00341112], [-0.00936695, -0.00673899], [-0.00240039, -0.00314409],
    [-0.01434128, -0.00585390], [0.00874225, 0.00295633], [0.00736213, -0.00328875],
    [0.00585082, -0.01232619]])


def wgn_case():
    true_param = P(10.0, 30.0, -0.15)
    initial_param = true_param + P(0.3, -0.5, 0.001)
    Tt = O.transform_new(true_param)
    Ti = O.transform_new(initial_param)
    dst = O.transform_apply_many(Tt, WGN_SRC) + WGN_NOISE
    return initial_param, Ti, WGN_SRC, dst


def test_weighted_gauss_newton_update():  # lib.rs:429-507
    initial_param, Ti, src, dst = wgn_case()
    assert len(src) == len(WGN_NOISE)
    rc, update = O.weighted_gauss_newton_update(Ti, src, dst)
    assert rc == O.OK
    Tu = O.transform_new(initial_param + update)
    e0 = O.error(Ti, src, dst)
    e1 = O.error(Tu, src, dst)
    assert e1 < e0 * 0.1
    Te, _ = O.estimate_transform(src, dst)
    e1 = O.error(Te, src, dst)
    assert e1 < e0 * 0.001


# the reference writes the literals 0.0, 0.1, ..., 1.0 (lib.rs:511-532, 555-576)
L_SHAPE_2D = np.array([[0.0, v] for v in (0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0)] +
                      [[v, 0.0] for v in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0)])


def test_icp_3dscan():  # lib.rs:509-551
    src = np.concatenate([L_SHAPE_2D, np.array([[2.0]] * 11 + [[1.0]] * 10)], axis=1)
    Tt = O.transform_new(P(0.01, 0.01, -0.02))
    dst = np.array([O.transform_xy(Tt, p) for p in src])
    noise = O.transform_new(P(0.05, 0.010, 0.010))
    Ti = O.transform_mul(noise, Tt)
    for kd in (False, True):
        rc, Tp, _, _ = O.icp_estimate(3, dst, src, Ti, 20, use_kdtree=kd)
        assert rc == O.OK
        for sp, dp_true in zip(src, dst):
            dp_pred = O.transform_xy(Tp, sp)
            assert norm(dp_pred - dp_true) < 1e-3


def test_icp_2dscan():  # lib.rs:553-595
    src = L_SHAPE_2D
    Tt = O.transform_new(P(0.01, 0.01, -0.02))
    dst = O.transform_apply_many(Tt, src)
    noise = O.transform_new(P(0.05, 0.010, 0.010))
    Ti = O.transform_mul(noise, Tt)
    for kd in (False, True):
        rc, Tp, _, _ = O.icp_estimate(2, dst, src, Ti, 20, use_kdtree=kd)
        assert rc == O.OK
        for sp, dp_true in zip(src, dst):
            dp_pred = O.transform_apply(Tp, sp)
            assert norm(dp_pred - dp_true) < 1e-3


# -------------------------------------------------------------- src/huber.rs tests --


def test_rho():  # huber.rs:33-37
    rho = O.lib().orc_huber_rho
    assert rho(0.1 * 0.1, 0.1) == 0.1 * 0.1
    assert rho(0.101 * 0.101, 0.1) == 2.0 * 0.1 * 0.101 - 0.1 * 0.1
    assert rho(0.09 * 0.09, 0.1) == 0.09 * 0.09


def powi2(x):  # f64::powi(2) is x*x
    return x * x


def test_drho():  # huber.rs:40-70
    rho, drho = O.lib().orc_huber_rho, O.lib().orc_huber_drho
    e1, e0, k = powi2(4.000 + 0.001), powi2(4.000), 4.0
    assert abs(drho(e0, k) - (rho(e1, k) - rho(e0, k)) / (e1 - e0)) < 1e-3
    e1, e0, k = powi2(0.10 + 0.01), powi2(0.10), 4.0
    assert (rho(e1, k) - rho(e0, k)) / (e1 - e0) == drho(e0, k)
    e1, e0, k = powi2(0.10 + 0.0001), powi2(0.10), 0.10
    assert abs(drho(e0, k) - (rho(e1, k) - rho(e0, k)) / (e1 - e0)) < 1e-3
    e1, e0, k = powi2(5.000 + 0.001), powi2(5.000), 4.0
    assert abs(drho(e0, k) - (rho(e1, k) - rho(e0, k)) / (e1 - e0)) < 1e-3
    e1, e0, k = powi2(10.000 + 0.001), powi2(10.000), 4.0
    assert abs(drho(e0, k) - (rho(e1, k) - rho(e0, k)) / (e1 - e0)) < 1e-3


# -------------------------------------------------------------- src/stats.rs tests --


def test_mutable_median():  # stats.rs:69-90
    assert O.median([-9., -6., -4., -1., -6., 5., 8., 5., 5., 4.]) == (O.OK, 1.5)
    assert O.median([15., 34., 26., -76., -19., 25., 93., -99., -52., 12., 6., -70., 59., 78., 69., -6.,
                     -33., 2., -27.]) == (O.OK, 6.0)
    assert O.median([-19., 38., -45., 35., 36., 68., 26., -27., 52., 41.]) == (O.OK, 35.5)
    assert O.median([])[0] == O.NONE
    assert O.median([50.]) == (O.OK, 50.)
    assert O.median([10., 11.]) == (O.OK, 10.5)


def test_mutable_mad():  # stats.rs:93-102
    assert O.mad([16., -16., -1., 8., -9., 4., -3., 17., 3., -7., 11., -1.]) == (O.OK, 7.5)
    assert O.mad([22., 1., -9., -35., -29., -40., -50., -45., 4.]) == (O.OK, 20.0)
    assert O.mad([-53., -36.]) == (O.OK, 8.5)


NORMAL_100 = [
    53.08322030, 60.78675339, 49.15066951, 60.1084452, 72.01118924, 50.04284213, 52.83008308,
    23.96785563, 35.51235652, 43.34002764, 46.38651612, 44.12070351, 44.17867909, 50.98783254,
    44.21536288, 70.17936403, 48.84330478, 51.58408135, 49.24294933, 56.12224494, 54.15417157,
    58.76714865, 52.41643234, 48.81350439, 42.27442158, 59.08548828, 40.58795014, 46.05835979,
    61.0659236, 42.13175052, 52.97283003, 39.46370987, 52.00781300, 39.87764594, 47.84026502,
    54.53531844, 39.01183939, 43.53705067, 49.98653523, 60.42712260, 28.35086716, 44.39726399,
    43.61557885, 63.29068847, 41.32778574, 51.68182699, 50.74441992, 47.43624869, 47.06234944,
    55.33085634, 60.17426330, 53.26886399, 35.19542111, 56.83354548, 31.65618383, 40.08374876,
    50.15219264, 44.44536522, 48.30516233, 65.41939507, 45.55690819, 55.68155501, 59.05170952,
    45.17456062, 57.80619559, 66.05259975, 46.00590789, 32.26217060, 55.38730483, 45.73005193,
    45.71435278, 55.95660079, 55.62156553, 48.26003878, 31.28428240, 55.10124146, 59.18713651,
    49.60689857, 61.96388754, 30.00022221, 60.35928071, 62.12555809, 46.91947312, 54.29469848,
    37.60662842, 47.93826864, 57.90926871, 44.36232644, 41.34588408, 42.27201939, 51.36323355,
    39.08440872, 53.04656841, 54.82787657, 46.40165516, 25.48827449, 56.49926944, 42.09583490,
    33.46258109, 43.52375750]


def test_mutable_standard_deviation():  # stats.rs:105-136
    rc, s = O.standard_deviation(NORMAL_100)
    assert rc == O.OK
    assert abs(s - 9.427146244705945) < 0.5


MEASUREMENTS_30 = np.array([
    [53.72201757, 52.99126564], [47.10884813, 53.59975516], [39.39661665, 61.08762518],
    [62.81692917, 54.56765183], [39.26208329, 45.65102341], [50.86473295, 44.72763481],
    [39.28791948, 34.88506328], [55.25576933, 39.59323902], [36.75721579, 57.17795218],
    [30.13909168, 64.76416708], [44.81493956, 54.94041174], [53.88324537, 60.4374775],
    [47.88396982, 66.59441293], [64.42865488, 40.9932948], [44.81265264, 50.45413795],
    [53.19558104, 28.24225202], [55.95984582, 65.33672375], [59.05920996, 27.61279324],
    [46.8073715, 30.79477285], [39.59866249, 45.6226116], [49.15739909, 55.53557656],
    [43.24838042, 43.95231977], [54.78299967, 40.5593425], [41.9153867, 55.54639181],
    [52.18015184, 46.38912455], [29.59992903, 46.32180761], [75.51275641, 57.73265648],
    [61.78180837, 54.48655747], [72.17828583, 66.37805296], [41.72995451, 50.9864875]])


def test_calc_stddevs():  # stats.rs:139-180
    rc, s = O.calc_stddevs(MEASUREMENTS_30)
    assert rc == O.OK
    assert abs(s[0] - 10.88547151) < 1.0
    assert abs(s[1] - 10.75361579) < 1.0


def test_median_nan_is_the_panic_case():  # stats.rs:12 partial_cmp().unwrap()
    assert O.median([1.0, float("nan"), 2.0])[0] == O.NAN


# ------------------------------------------------------------- src/linalg.rs tests --


def test_inverse3x3():  # linalg.rs:37-72
    I = np.eye(3)
    m = np.array([[-3.64867356, 0.11236464, -7.60555263], [-3.56881707, -9.77855129, 0.50475873],
                  [-9.34728378, 0.25373179, -7.55422161]])
    rc, inv = O.inverse3x3(m)
    assert rc == O.OK
    assert norm(inv @ m - I) < 1e-14
    assert O.inverse3x3(np.zeros((3, 3)))[0] == O.NONE
    m = np.array([[3.0, 1.0, 2.0], [6.0, 2.0, 4.0], [9.0, 9.0, 7.0]])
    assert O.inverse3x3(m)[0] == O.NONE
    m = np.array([[3.00792510e-38, -1.97985750e-45, 3.61627897e-44],
                  [7.09699991e-49, -3.08764937e-49, -8.31427092e-41],
                  [2.03723891e-42, -3.84594910e-42, 1.00872600e-40]])
    rc, inv = O.inverse3x3(m)
    assert rc == O.OK
    assert norm(inv @ m - I) < 1e-14


# ---------------------------------------------------------------- src/se2.rs tests --


def se2_exp(param):
    import ctypes as C

    out = np.zeros(9)
    p = np.array(param, dtype=np.float64)
    O.lib().orc_se2_exp(p.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out.reshape(3, 3)


def se2_log(m):
    import ctypes as C

    out = np.zeros(3)
    m = np.ascontiguousarray(m, dtype=np.float64)
    O.lib().orc_se2_log(m.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def test_se2_exp():  # se2.rs:85-142
    t = se2_exp([-0.29638466, -0.15797957, -0.89885138])
    e = np.array([[0.6225093, 0.7826124, -0.32440305], [-0.7826124, 0.6225093, -0.01307704], [0., 0., 1.]])
    assert norm(t - e) < 1e-6
    t = se2_exp([-0.24295876, 0.95847196, 0.91052553])
    e = np.array([[0.61333076, -0.78982617, -0.61778258], [0.78982617, 0.61333076, 0.72824049], [0., 0., 1.]])
    assert norm(t - e) < 1e-6
    t = se2_exp([10., -20., 0.])
    e = np.array([[1., 0., 10.], [0., 1., -20.], [0., 0., 1.]])
    assert norm(t - e) < 1e-6


def test_se2_log():  # se2.rs:145-200
    m = np.array([[-7.18473159e-02, 9.97415642e-01, 1.98003686e+00],
                  [-9.97415642e-01, -7.18473159e-02, -1.67935601e+00],
                  [0.00000000e+00, 1.11022302e-16, 1.00000000e+00]])
    assert norm(se2_log(m) - P(2.89271776, 0.34275002, -1.6427056)) < 1e-6
    m = np.array([[-1.0, 0.0, -1.90985932e+00], [0.0, -1.0, -6.36619772e-01], [0.0, 0.0, 1.0]])
    assert norm(se2_log(m) - P(-1., 3., PI)) < 1e-6
    m = np.array([[1., 0., -1.], [0., 1., 3.], [0., 0., 1.]])
    assert norm(se2_log(m) - P(-1., 3., 0.)) < 1e-6


def test_se2_get_rt():  # se2.rs:203-221
    import ctypes as C

    m = np.array([[0.6225093, 0.7826124, -0.32440305], [-0.7826124, 0.6225093, -0.01307704], [0., 0., 1.]])
    rot = np.zeros(4)
    t = np.zeros(2)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    O.lib().orc_se2_get_rt(dp(np.ascontiguousarray(m)), dp(rot), dp(t))
    assert np.array_equal(rot.reshape(2, 2), np.array([[0.6225093, 0.7826124], [-0.7826124, 0.6225093]]))
    assert np.array_equal(t, P(-0.32440305, -0.01307704))


# ---------------------------------------------------------------- src/so2.rs tests --


def so2_exp(theta):
    import ctypes as C

    m = np.zeros(4)
    O.lib().orc_so2_exp(theta, m.ctypes.data_as(C.POINTER(C.c_double)))
    return m  # column-major: (0,0), (1,0), (0,1), (1,1)


def test_so2_exp():  # so2.rs:39-48
    theta = 0.3
    m = so2_exp(theta)
    assert m[0] == math.cos(theta)
    assert m[2] == -math.sin(theta)
    assert m[1] == math.sin(theta)
    assert m[3] == math.cos(theta)


def test_so2_log():  # so2.rs:51-67
    import ctypes as C

    for f in (0.3, 0.8, -0.7, -0.1):
        theta = f * PI
        m = so2_exp(theta)
        assert abs(O.lib().orc_so2_log(m.ctypes.data_as(C.POINTER(C.c_double))) - theta) < 1e-6


# ---------------------------------------------------------- src/transform.rs tests --


def from_rt(theta, t):
    m = so2_exp(theta)
    return O.Pose(m[0], m[1], m[2], m[3], t[0], t[1])


def test_transform():  # transform.rs:62-70
    T = from_rt(PI / 2, (3., 6.))
    assert norm(O.transform_apply(T, P(4., 2.)) - P(-2. + 3., 4. + 6.)) < 1e-8


def test_inverse():  # transform.rs:73-80
    T = O.transform_inverse(from_rt(PI / 2, (3., 6.)))
    assert norm(O.transform_apply(T, P(-2. + 3., 4. + 6.)) - P(4., 2.)) < 1e-8


def test_mul():  # transform.rs:83-96
    T1 = O.transform_inverse(from_rt(PI / 4, (2., 1.)))
    T2 = O.transform_inverse(from_rt(PI / 2, (5., 3.)))
    x = P(-5., 6.)
    pa = O.transform_apply(T1, O.transform_apply(T2, x))
    pb = O.transform_apply(O.transform_mul(T1, T2), x)
    assert norm(pa - pb) < 1e-8


# --------------------------------------------- oracle-internal consistency (not KATs) --


def test_kdtree_equals_brute_force_including_ties():
    rng = np.random.default_rng(7)
    for dim in (2, 3):
        dst = rng.integers(-6, 6, size=(400, dim)).astype(np.float64)  # many exact ties + duplicates
        q = rng.integers(-7, 7, size=(300, dim)).astype(np.float64) + 0.5 * rng.integers(0, 2, size=(300, dim))
        rc, ib = O.nn_brute(dst, q)
        assert rc == O.OK
        rc, ik = O.KdTree(dst).search(q)
        assert rc == O.OK
        assert np.array_equal(ib, ik)
        dst = rng.normal(size=(3000, dim))
        q = rng.normal(size=(2000, dim))
        assert np.array_equal(O.nn_brute(dst, q)[1], O.KdTree(dst).search(q)[1])


def test_median_matches_sorting():
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 10, 101, 1000, 4097):
        v = rng.normal(size=n)
        v[rng.integers(0, n, size=n // 3)] = 0.25  # duplicates
        s = np.sort(v)
        want = s[n // 2] if n % 2 else (s[n // 2 - 1] + s[n // 2]) / 2.0
        assert O.median(v) == (O.OK, want)


def test_tree_sum_variant_close_to_left_fold():
    _, Ti, src, dst = wgn_case()
    rc0, d0 = O.weighted_gauss_newton_update(Ti, src, dst)
    rc1, d1, err = O.weighted_gauss_newton_update_tree(Ti, src, dst, 4, 64)
    assert rc0 == rc1 == O.OK
    assert np.allclose(d0, d1, rtol=1e-12, atol=0)
    assert abs(err - O.huber_error(Ti, src, dst)) <= 1e-12 * abs(err)


def test_empty_dst_is_the_panic_case():  # lib.rs:122,165 index.unwrap()
    rc, _, _, _ = O.icp_estimate(2, np.zeros((0, 2)), np.array([[1.0, 2.0]]), O.transform_identity(), 1)
    assert rc == O.EMPTY_DST
