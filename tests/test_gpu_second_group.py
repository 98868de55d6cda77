"""GPU parity tests, second group of iterations: SFISTA, DavisYin, LiLin, DRLS, AFBA / VuCondat / ChambollePock and the
operators they need (LeastSquares.prox_, SqrNormL2, SquaredDistance.prox_, Quadratic, Conjugate).  Each test mirrors
a reference test (known answer + iteration bound) on the device AND compares the device iterates with the CPU oracle
(oracle/proxgrad_oracle_ext.py) on the same inputs.

Tolerances: iterate sequences 2e-4 * max(1, ||.||_inf) in Float32 over <= 40 iterations (these methods chain
several rounded AXPBYs per step; the oracle rounds in a different grouping), 1e-10 in Float64; iteration counts equal
to the oracle's +- 1 in Float32 (a stop test within rounding of the tolerance) and equal in Float64.
"""
import numpy as np
import pytest

import reference_vectors as rv
from oracle import proxgrad_oracle as o
from oracle import proxgrad_oracle_ext as ox

pytestmark = pytest.mark.gpu
DTYPES = [np.float32, np.float64]


@pytest.fixture(scope="module")
def pa():
    import proximalalgorithms.jl_amd as pa

    pa.get_context()  # raises loudly when the HIP library / device is missing
    return pa


def seq_tol(dtype):
    return 2e-4 if np.dtype(dtype) == np.float32 else 1e-10


def lasso_small(dtype):
    A = np.asfortranarray(rv.LASSO_SMALL_A.astype(dtype))
    b = rv.LASSO_SMALL_B.astype(dtype)
    R = np.dtype(dtype).type
    lam = R(0.1) * R(np.max(np.abs(A.T @ b)))
    Lf = R(np.linalg.norm(A, 2) ** 2)
    return A, b, lam, Lf


def close(got, ref, dtype, scale=1.0):
    ref = np.asarray(ref)
    return np.max(np.abs(got - ref)) <= scale * seq_tol(dtype) * max(1.0, float(np.max(np.abs(ref))))


def same_count(it, it_ref, dtype):
    return it == it_ref if np.dtype(dtype) == np.float64 else abs(it - it_ref) <= 1


# ------------------------------------------------------------------------------------------------
# operators
# ------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n", [(4, 5), (40, 12), (30, 200), (300, 64)])
def test_least_squares_prox_matches_direct_solve(pa, dtype, m, n):
    rng = np.random.default_rng(m + n)
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    b, x = rng.standard_normal(m).astype(dtype), rng.standard_normal(n).astype(dtype)
    for lam_ls, gamma in ((1.0, 0.7), (0.5, 3.0)):
        f = pa.LeastSquares(A, b, lam=lam_ls)
        y = pa.HIPVector.empty(n, dtype)
        fy = f.prox_(y, pa.HIPVector.from_numpy(x), gamma)
        A64, b64, x64 = A.astype(np.float64), b.astype(np.float64), x.astype(np.float64)
        ref = np.linalg.solve(lam_ls * A64.T @ A64 + np.eye(n) / gamma, lam_ls * A64.T @ b64 + x64 / gamma)
        tol = 5e-5 if dtype == np.float32 else 1e-11
        assert np.max(np.abs(y.numpy() - ref)) <= tol * max(1.0, np.max(np.abs(ref)))
        assert float(fy) == pytest.approx(lam_ls / 2 * np.sum((A64 @ ref - b64) ** 2), rel=10 * tol, abs=tol)
        # the restatement's prox agrees as well
        yo, _ = o.LeastSquares(A, b, lam_ls).prox(x, dtype(gamma))
        assert np.max(np.abs(y.numpy() - yo)) <= 20 * tol * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("dtype", DTYPES)
def test_small_operators_match_oracle(pa, dtype):
    rng = np.random.default_rng(7)
    n = 1000
    x, b = rng.standard_normal(n).astype(dtype), rng.standard_normal(n).astype(dtype)
    xd = pa.HIPVector.from_numpy(x)
    y = xd.similar()
    eps = np.finfo(dtype).eps
    for gamma in (0.3, 2.5):
        for dev, ora in ((pa.SqrNormL2(1.7), ox.SqrNormL2(1.7)), (pa.SquaredDistance(b, 0.6), ox.SqrDistance(b, 0.6)),
                         (pa.Conjugate(pa.NormL1(0.4)), ox.Conjugate(o.NormL1(0.4))),
                         (pa.Conjugate(pa.SquaredDistance(b)), ox.Conjugate(ox.SqrDistance(b)))):
            v = dev.prox_(y, xd, gamma)
            yo, vo = ora.prox(x, dtype(gamma))
            assert np.max(np.abs(y.numpy() - yo)) <= 8 * eps * max(1.0, np.max(np.abs(yo)))
            assert float(v) == pytest.approx(float(vo), rel=2e-5 if dtype == np.float32 else 1e-12, abs=1e-4 if dtype == np.float32 else 1e-10)
    for dev, ora in ((pa.SqrNormL2(1.7), ox.SqrNormL2(1.7)), (pa.SquaredDistance(b, 0.6), ox.SqrDistance(b, 0.6))):
        v, g = dev.value_and_gradient(xd)
        vo, go = ora.value_and_gradient(x)
        assert np.max(np.abs(g.numpy() - go)) <= 8 * eps * max(1.0, np.max(np.abs(go)))
        assert float(v) == pytest.approx(float(vo), rel=2e-5 if dtype == np.float32 else 1e-12)
    Q = rng.standard_normal((60, 60)).astype(dtype)
    Q = np.asfortranarray(Q + Q.T)
    q, z = rng.standard_normal(60).astype(dtype), rng.standard_normal(60).astype(dtype)
    v, g = pa.Quadratic(Q, q).value_and_gradient(pa.HIPVector.from_numpy(z))
    vo, go = o.Quadratic(Q, q).value_and_gradient(z)
    assert np.max(np.abs(g.numpy() - go)) <= 200 * eps * np.max(np.abs(go))
    assert float(v) == pytest.approx(float(vo), rel=1e-4 if dtype == np.float32 else 1e-12)
    assert pa.convex_conjugate(pa.IndZero()).__class__ is pa.Zero
    assert pa.is_convex(pa.NormL1(1.0)) and pa.is_generalized_quadratic(pa.LeastSquares(Q, q)) and not pa.is_convex(object())


# ------------------------------------------------------------------------------------------------
# SFISTA (test_lasso_small.jl:274-283, test_lasso_small_strongly_convex.jl:56-65)
# ------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("dtype", DTYPES)
def test_sfista_pins_and_oracle(pa, dtype):
    A, b, lam, Lf = lasso_small(dtype)
    x0 = np.zeros(5, dtype)
    y, it = pa.SFISTA(tol=10 * rv.LASSO_SMALL_TOL)(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam), Lf=Lf)
    assert isinstance(y, np.ndarray) and y.dtype == dtype and np.all(x0 == 0)
    assert np.max(np.abs(y - rv.LASSO_SMALL_XSTAR.astype(dtype))) <= 10 * rv.LASSO_SMALL_TOL
    assert it < rv.LASSO_SMALL_BOUNDS_EXT["sfista"]
    yo, ito = ox.sfista(tol=10 * rv.LASSO_SMALL_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=Lf)
    assert same_count(it, ito, dtype) and close(y, yo, dtype, 5)
    # strongly convex instance, mf > 0
    A, b, lam, x0 = rv.strongly_convex_problem(dtype)
    y, it = pa.SFISTA(tol=rv.SC_TOL)(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam), Lf=rv.SC_LF, mf=rv.SC_MF)
    assert np.linalg.norm(y - rv.SC_XSTAR.astype(dtype)) <= rv.SC_TOL and it < rv.SC_BOUNDS_EXT["sfista"]
    _, ito = ox.sfista(tol=rv.SC_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=rv.SC_LF, mf=rv.SC_MF)
    assert same_count(it, ito, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_sfista_iterates_match_oracle(pa, dtype):
    rng = np.random.default_rng(21)
    m, n = 120, 300
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    b = rng.standard_normal(m).astype(dtype)
    R = np.dtype(dtype).type
    lam = R(0.1) * R(np.max(np.abs(A.T @ b)))
    Lf = R(np.linalg.norm(A, 2) ** 2)
    x0 = (0.1 * rng.standard_normal(n)).astype(dtype)
    dev = iter(pa.SFISTAIteration(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam), Lf=Lf, mf=R(0.05)))
    ora = iter(ox.SFISTAIteration(x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=Lf, mf=R(0.05)))
    for k in range(30):
        sd, so = next(dev), next(ora)
        assert close(sd.y.numpy(), so.y, dtype), k
        assert close(sd.x.numpy(), so.x, dtype, 5), k
        assert float(sd.A) == pytest.approx(float(so.A), rel=1e-6)


# ------------------------------------------------------------------------------------------------
# DavisYin (test_elasticnet.jl:31-56)
# ------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("dtype", DTYPES)
def test_davis_yin_pins_and_oracle(pa, dtype):
    A, b, _, Lf = lasso_small(dtype)
    R = np.dtype(dtype).type
    for x0 in (np.zeros(5, dtype), np.random.default_rng(3).standard_normal(5).astype(dtype)):
        x0b = x0.copy()
        x, it = pa.DavisYin(tol=R(1e-6))(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(R(1)), h=pa.SqrNormL2(R(1)), Lf=Lf)
        assert x.dtype == dtype and np.array_equal(x0, x0b)
        assert np.max(np.abs(x - rv.ELASTICNET_XSTAR.astype(dtype))) <= rv.ELASTICNET_DYS["x_tol"]
        xo, ito = ox.davis_yin(tol=R(1e-6), x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(R(1)), h=ox.SqrNormL2(R(1)), Lf=Lf)
        if not x0.any():
            assert it <= rv.ELASTICNET_DYS["it"]
        assert abs(it - ito) <= (3 if dtype == np.float32 else 0) and close(x, xo, dtype)
    with pytest.raises(ValueError):
        pa.DavisYinIteration(x0=np.zeros(5, dtype), f=pa.LeastSquares(A, b))  # neither Lf nor gamma (davis_yin.jl:48-49)
    dev = iter(pa.DavisYinIteration(x0=np.ones(5, dtype), f=pa.LeastSquares(A, b), g=pa.NormL1(R(1)), h=pa.IndBox(-0.5, 0.5),
                                    gamma=R(0.9) / Lf, lam=R(1.3)))
    ora = iter(ox.DavisYinIteration(x0=np.ones(5, dtype), f=o.LeastSquares(A, b), g=o.NormL1(R(1)), h=o.IndBox(-0.5, 0.5),
                                    gamma=R(0.9) / Lf, lam=R(1.3)))
    for k in range(40):
        sd, so = next(dev), next(ora)
        assert close(sd.z.numpy(), so.z, dtype) and close(sd.xh.numpy(), so.xh, dtype), k


# ------------------------------------------------------------------------------------------------
# LiLin (test_nonconvex_qp.jl:58-66 and :125-133)
# ------------------------------------------------------------------------------------------------


def test_lilin_nonconvex_qp(pa):
    Q, q = np.diag(rv.NCQP_Q_DIAG), rv.NCQP_Q_VEC
    gamma = 0.95 / np.max(rv.NCQP_Q_DIAG)
    x0 = np.zeros(2)
    x, it = pa.LiLin(gamma=gamma, tol=rv.NCQP_TOL)(x0=x0, f=pa.Quadratic(Q, q), g=pa.IndBox(-1.0, 1.0))
    z = np.minimum(1.0, np.maximum(-1.0, x - gamma * (Q @ x + q)))
    assert np.max(np.abs(x - z)) / gamma <= rv.NCQP_TOL and np.all(x0 == 0)
    xo, ito = ox.li_lin(tol=rv.NCQP_TOL, x0=x0, f=o.Quadratic(Q, q), g=o.IndBox(-1.0, 1.0), gamma=gamma)
    assert it == ito and np.max(np.abs(x - xo)) <= 1e-12
    # "small" instances of the same file: random symmetric Q with a negative eigenvalue (recipe restated with numpy's
    # generator -- the Julia RNG stream cannot be reproduced here; the property asserted is the reference's)
    for k in range(1, 4):
        rng = np.random.default_rng(k)
        n = 100
        U, _ = np.linalg.qr(rng.standard_normal((n, n)))
        eigs = np.concatenate([-rng.random(n // 2) * 0.1, rng.random(n - n // 2)])
        Q = np.asfortranarray((U * eigs[None, :]) @ U.T)
        Q = (Q + Q.T) / 2
        q = rng.standard_normal(n)
        gamma = 0.95 / np.max(np.abs(eigs))
        x0 = np.zeros(n)
        x, it = pa.LiLin(gamma=gamma, tol=1e-4, maxit=20000)(x0=x0, f=pa.Quadratic(Q, q), g=pa.IndBox(-1.0, 1.0))
        z = np.minimum(1.0, np.maximum(-1.0, x - gamma * (Q @ x + q)))
        assert np.max(np.abs(x - z)) / gamma <= 1e-4
        dev = iter(pa.LiLinIteration(x0=x0, f=pa.Quadratic(Q, q), g=pa.IndBox(-1.0, 1.0), gamma=gamma))
        ora = iter(ox.LiLinIteration(x0=x0, f=o.Quadratic(Q, q), g=o.IndBox(-1.0, 1.0), gamma=gamma))
        for j in range(40):
            sd, so = next(dev), next(ora)
            assert np.max(np.abs(sd.z.numpy() - so.z)) <= 1e-10, (k, j)
            assert float(sd.F_average) == pytest.approx(float(so.F_average), rel=1e-10, abs=1e-12)


@pytest.mark.parametrize("dtype", DTYPES)
def test_lilin_monitor_branch_matches_oracle(pa, dtype):
    """Both branches of li_lin.jl:106-125 (the monitor branch is forced by a large delta)."""
    A, b, lam, Lf = lasso_small(dtype)
    R = np.dtype(dtype).type
    x0 = np.ones(5, dtype)
    kw = dict(gamma=R(0.9) / Lf, delta=R(1e3))
    di = pa.LiLinIteration(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam), **kw)
    oi = ox.LiLinIteration(x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), **kw)
    dev, ora = iter(di), iter(oi)
    for k in range(25):
        sd, so = next(dev), next(ora)
        assert close(sd.z.numpy(), so.z, dtype, 5) and close(sd.y.numpy(), so.y, dtype, 5), k
    assert di.monitor_branch_taken == oi.monitor_branch_taken > 0
    # f = LeastSquares on a device matrix: one read of A per iteration, one more whenever the monitor branch runs
    assert di.counters["a_passes"] == 2 + 24 + di.monitor_branch_taken
    dp = pa.LiLinIteration(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam), single_sweep=False, **kw)
    for k, (s1, s2) in enumerate(zip(pa.LiLinIteration(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam), **kw), dp)):
        if k >= 25:
            break
        assert close(s1.z.numpy(), s2.z.numpy(), dtype, 5), k


# ------------------------------------------------------------------------------------------------
# DouglasRachford with f = LeastSquares on the device (test_lasso_small.jl:205-214) and DRLS (:216-231,
# test_lasso_small_strongly_convex.jl:146-153, test_equivalence.jl:14-49)
# ------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("dtype", DTYPES)
def test_douglas_rachford_lasso_on_device(pa, dtype):
    A, b, lam, Lf = lasso_small(dtype)
    R = np.dtype(dtype).type
    x0 = np.zeros(5, dtype)
    gamma = R(10) / R(np.linalg.norm(A, 2) ** 2)
    y, it = pa.DouglasRachford(gamma=gamma, tol=rv.LASSO_SMALL_TOL)(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam))
    assert y.dtype == dtype and np.all(x0 == 0)
    assert np.max(np.abs(y - rv.LASSO_SMALL_XSTAR.astype(dtype))) <= rv.LASSO_SMALL_TOL
    assert it < rv.LASSO_SMALL_BOUNDS_EXT["dr"]
    _, ito = o.douglas_rachford(gamma=gamma, tol=rv.LASSO_SMALL_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam))
    assert same_count(it, ito, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["lbfgs", "nesterov_fixed", "nesterov_simple"])
def test_drls_lasso_pins_and_oracle(pa, dtype, kind):
    A, b, lam, Lf = lasso_small(dtype)
    x0 = np.zeros(5, dtype)
    directions = {"lbfgs": pa.LBFGS(5), "nesterov_fixed": pa.NesterovExtrapolation(pa.FixedNesterovSequence),
                  "nesterov_simple": pa.NesterovExtrapolation(pa.SimpleNesterovSequence)}[kind]
    z, it = pa.DRLS(tol=10 * rv.LASSO_SMALL_TOL, directions=directions)(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam),
                                                                         Lf=Lf)
    assert z.dtype == dtype and np.all(x0 == 0)
    assert np.max(np.abs(z - rv.LASSO_SMALL_XSTAR.astype(dtype))) <= 10 * rv.LASSO_SMALL_TOL
    assert it < rv.LASSO_SMALL_BOUNDS_EXT["drls_" + kind]
    zo, ito = ox.drls(tol=10 * rv.LASSO_SMALL_TOL, directions=kind, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=Lf)
    # Float32: the envelope comparisons of the line search fall either way within rounding (the restatement's Cholesky
    # solve and the device's pre-inverted system round differently), which shifts the count by several iterations; both
    # runs satisfy the reference's pin above.  Float64: identical counts.
    assert it == ito if dtype == np.float64 else (ito < rv.LASSO_SMALL_BOUNDS_EXT["drls_" + kind] and abs(it - ito) <= 12)
    if dtype == np.float64:
        assert close(z, zo, dtype, 10)


@pytest.mark.parametrize("dtype", DTYPES)
def test_drls_strongly_convex_and_dr_equivalence(pa, dtype):
    A, b, lam, x0 = rv.strongly_convex_problem(dtype)
    v, it = pa.DRLS(tol=rv.SC_TOL)(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam), mf=rv.SC_MF)
    assert np.max(np.abs(v - rv.SC_XSTAR.astype(dtype))) <= rv.SC_TOL and it < rv.SC_BOUNDS_EXT["drls"]
    # test_equivalence.jl:14-49: DRLS without acceleration and c = -Inf is plain DouglasRachford
    A, b, lam, Lf = lasso_small(dtype)
    R = np.dtype(dtype).type
    f, g = pa.LeastSquares(A, b), pa.NormL1(lam)
    gamma = R(10) / R(np.linalg.norm(A, 2) ** 2)
    x0 = np.zeros(5, dtype)
    dr = iter(pa.DouglasRachfordIteration(f=f, g=g, x0=x0, gamma=gamma))
    dl = iter(pa.DRLSIteration(f=pa.LeastSquares(A, b), g=g, x0=x0, gamma=gamma, lam=R(1), c=-np.inf, max_backtracks=1,
                               directions=pa.NoAcceleration()))
    for _ in range(10):
        a, bb = next(dr), next(dl)
        assert np.allclose(a.x.numpy(), bb.xbar.numpy(), rtol=np.sqrt(np.finfo(dtype).eps), atol=0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_drls_box_qp_matches_oracle(pa, dtype):
    """separable-quadratic f (the DouglasRachford config-3 operator pair), general line-search branch statistics"""
    rng = np.random.default_rng(9)
    n = 5000
    d = (0.5 + rng.random(n)).astype(dtype)
    q = rng.standard_normal(n).astype(dtype)
    x0 = np.zeros(n, dtype)
    kw = dict(Lf=float(np.max(d)), mf=float(np.min(d)))
    z, it = pa.DRLS(tol=1e-5 if dtype == np.float32 else 1e-9)(x0=x0, f=pa.SeparableQuadratic(d, q), g=pa.IndBox(-0.5, 0.5), **kw)
    zo, ito = ox.drls(tol=1e-5 if dtype == np.float32 else 1e-9, x0=x0, f=o.SeparableQuadratic(d, q), g=o.IndBox(-0.5, 0.5), **kw)
    exact = np.clip(-q.astype(np.float64) / d.astype(np.float64), -0.5, 0.5)
    assert np.max(np.abs(z - exact)) <= (1e-4 if dtype == np.float32 else 1e-8)
    assert abs(it - ito) <= 2 and np.max(np.abs(z - zo)) <= (1e-4 if dtype == np.float32 else 1e-8)


# ------------------------------------------------------------------------------------------------
# AFBA / VuCondat / ChambollePock (test_lasso_small.jl:233-272, test_elasticnet.jl:58-120)
# ------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("dtype", DTYPES)
def test_afba_lasso_pins_and_oracle(pa, dtype):
    A, b, lam, Lf = lasso_small(dtype)
    R = np.dtype(dtype).type
    x0 = np.zeros(5, dtype)
    xs = rv.LASSO_SMALL_XSTAR.astype(dtype)
    cases = [
        (dict(y0=np.zeros(5, dtype), f=pa.LeastSquares(A, b), g=pa.NormL1(lam), beta_f=Lf),
         dict(y0=np.zeros(5, dtype), f=o.LeastSquares(A, b), g=o.NormL1(lam), beta_f=Lf), "afba_f_g"),
        (dict(y0=np.zeros(5, dtype), f=pa.LeastSquares(A, b), h=pa.NormL1(lam), beta_f=Lf),
         dict(y0=np.zeros(5, dtype), f=o.LeastSquares(A, b), h=o.NormL1(lam), beta_f=Lf), "afba_f_h"),
        (dict(y0=np.zeros(4, dtype), h=pa.SquaredDistance(b), L=A, g=pa.NormL1(lam)),
         dict(y0=np.zeros(4, dtype), h=ox.SqrDistance(b), L=A, g=o.NormL1(lam)), "afba_h_L_g"),
    ]
    for kd, ko, key in cases:
        (x, y), it = pa.AFBA(theta=1, mu=1, tol=R(1e-6))(x0=x0, **kd)
        assert x.dtype == dtype and y.dtype == dtype and np.all(x0 == 0)
        assert np.max(np.abs(x - xs)) <= 1e-4 and it <= rv.LASSO_SMALL_BOUNDS_EXT[key], key
        (xo, yo), ito = ox.afba(theta=1, mu=1, tol=R(1e-6), x0=x0, **ko)
        # Float32 with tol = 1e-6: the fixed-point residual sits at the rounding floor when the rule fires
        assert abs(it - ito) <= (max(3, ito // 10) if dtype == np.float32 else 0), key
        assert close(x, xo, dtype) and close(y, yo, dtype), key


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("theta,mu,maxit", rv.ELASTICNET_AFBA)
def test_afba_elasticnet_variants(pa, dtype, theta, mu, maxit):
    A, b, _, _ = lasso_small(dtype)
    R = np.dtype(dtype).type
    rng = np.random.default_rng(5)
    for x0, y0 in ((np.zeros(5, dtype), np.zeros(4, dtype)),
                   (rng.standard_normal(5).astype(dtype), rng.standard_normal(4).astype(dtype))):
        (x, y), it = pa.AFBA(theta=theta, mu=mu, tol=R(1e-6))(x0=x0, y0=y0, f=pa.SqrNormL2(R(1)), g=pa.NormL1(R(1)),
                                                               h=pa.SquaredDistance(b), L=A, beta_f=1)
        assert np.max(np.abs(x - rv.ELASTICNET_XSTAR.astype(dtype))) <= 1e-4
        if not x0.any():
            assert it <= maxit
        (xo, yo), ito = ox.afba(theta=theta, mu=mu, tol=R(1e-6), x0=x0, y0=y0, f=ox.SqrNormL2(R(1)), g=o.NormL1(R(1)),
                                h=ox.SqrDistance(b), L=A, beta_f=1)
        assert abs(it - ito) <= (max(3, ito // 10) if dtype == np.float32 else 0)
        assert close(x, xo, dtype) and close(y, yo, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_vu_condat_and_chambolle_pock_match_oracle(pa, dtype):
    rng = np.random.default_rng(13)
    m, n = 80, 150
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    b = rng.standard_normal(m).astype(dtype)
    R = np.dtype(dtype).type
    lam = R(0.1) * R(np.max(np.abs(A.T @ b)))
    x0, y0 = np.zeros(n, dtype), np.zeros(m, dtype)
    tol = R(1e-4 if dtype == np.float32 else 1e-8)
    (x, y), it = pa.ChambollePock(tol=tol)(x0=x0, y0=y0, g=pa.NormL1(lam), h=pa.SquaredDistance(b), L=A)
    (xo, yo), ito = ox.chambolle_pock(tol=tol, x0=x0, y0=y0, g=o.NormL1(lam), h=ox.SqrDistance(b), L=A)
    assert abs(it - ito) <= (max(3, ito // 50) if dtype == np.float32 else 0) and close(x, xo, dtype, 5) and close(y, yo, dtype, 5)
    # Vu-Condat with a smooth term: elastic-net objective
    (x, y), it = pa.VuCondat(tol=tol)(x0=x0, y0=y0, f=pa.SqrNormL2(R(0.5)), beta_f=0.5, g=pa.NormL1(lam),
                                       h=pa.SquaredDistance(b), L=A)
    (xo, yo), ito = ox.vu_condat(tol=tol, x0=x0, y0=y0, f=ox.SqrNormL2(R(0.5)), beta_f=0.5, g=o.NormL1(lam),
                                 h=ox.SqrDistance(b), L=A)
    assert abs(it - ito) <= (max(3, ito // 50) if dtype == np.float32 else 0) and close(x, xo, dtype, 5) and close(y, yo, dtype, 5)
    # KKT of the elastic net at the answer: 0 in 0.5 x + A'(Ax - b) + lam d|x|
    x64 = x.astype(np.float64)
    grad = 0.5 * x64 + A.astype(np.float64).T @ (A.astype(np.float64) @ x64 - b)
    viol = np.where(x64 != 0, np.abs(grad + float(lam) * np.sign(x64)), np.maximum(np.abs(grad) - float(lam), 0))
    assert np.max(viol) <= (1e-1 if dtype == np.float32 else 1e-4)  # FPR <= tol bounds the KKT residual times the step
    with pytest.raises(ValueError):
        pa.AFBAIteration(x0=x0, y0=y0, f=pa.SqrNormL2(1.0))  # beta_f must come with f (primal_dual.jl:96)
    with pytest.raises(ValueError):
        pa.AFBAIteration(x0=x0, y0=y0, theta=0.3, mu=0.7, h=pa.SquaredDistance(b), L=A)  # unsupported (theta, mu) (:414)


# ------------------------------------------------------------------------------------------------
# Anderson / Broyden operators (test/accel/test_anderson.jl, test_broyden.jl) and their use as directions
# ------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["anderson", "broyden"])
def test_anderson_and_broyden_on_device(pa, dtype, kind):
    R = np.dtype(dtype).type
    H, l = rv.ACCEL_H.astype(dtype), rv.ACCEL_L.astype(dtype)
    f = lambda x: R(np.dot(x, H @ x) / R(2) + np.dot(x, l))
    f_star = f(-np.linalg.solve(H, l))
    fq = pa.Quadratic(np.asfortranarray(H), np.zeros(5, dtype))  # H x through the device GEMV
    x = pa.HIPVector.zeros(5, dtype)
    tag = pa.AndersonAcceleration(5) if kind == "anderson" else pa.Broyden()
    acc = tag.initialize(x)
    acc_o = ox.AndersonAccelerationOperator(5, np.zeros(5, dtype)) if kind == "anderson" else ox.BroydenOperator(np.zeros(5, dtype))
    lv = pa.HIPVector.from_numpy(l)
    grad = lambda v: fq.value_and_gradient(v)[1].axpby_(1.0, fq.value_and_gradient(v)[1], 1.0, lv)
    g = grad(x)
    xo, go = np.zeros(5, dtype), l.copy()
    for it in range(rv.ACCEL_ITERS):
        d = acc * g
        do = acc_o * go
        if it < 3:  # same directions as the restatement while the (pseudo-inverted) memory is well conditioned
            assert np.max(np.abs(d.numpy() - do)) <= (2e-3 if dtype == np.float32 else 1e-9) * max(1.0, np.max(np.abs(do))), it
        x.axpby_(1.0, x, -1.0, d)
        g_prev, g = g, grad(x)
        md = d.similar().axpby_(-1.0, d)
        acc.update_(md, g.similar().axpby_(1.0, g, -1.0, g_prev))
        xo = xo - do
        go_prev, go = go, H @ xo + l
        acc_o.update(-do, go - go_prev)
    assert f(x.numpy()) <= f_star + (1 + abs(f_star)) * np.sqrt(np.finfo(dtype).eps)
    acc.reset_()
    assert np.array_equal((acc * x).numpy(), x.numpy())


@pytest.mark.parametrize("dtype", DTYPES)
def test_rank1_update_kernel(pa, dtype):
    import ctypes as C

    from proximalalgorithms.jl_amd import _lib

    rng = np.random.default_rng(2)
    for m, n in ((1, 1), (5, 5), (257, 33), (1000, 70)):
        A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype))
        u, w = rng.standard_normal(m).astype(dtype), rng.standard_normal(n).astype(dtype)
        Ad = pa.HIPMatrix.from_numpy(A)
        ud, wd = pa.HIPVector.from_numpy(u), pa.HIPVector.from_numpy(w)
        _lib.call("pg_mat_rank1_update", Ad.handle, 0.37, ud.vp, wd.vp)
        ref = A + (dtype(0.37) * w)[None, :] * u[:, None]
        assert np.max(np.abs(Ad.numpy() - ref)) <= 2 * np.finfo(dtype).eps * np.max(np.abs(ref))
        # padding rows stay zero: a product with the adjoint still matches
        r = rng.standard_normal(m).astype(dtype)
        g = Ad.mul_adjoint(pa.HIPVector.from_numpy(r)).numpy()
        assert np.max(np.abs(g - ref.T.astype(np.float64) @ r)) <= 1e-4 * max(1.0, np.max(np.abs(g)))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["broyden", "anderson"])
def test_drls_and_panoc_with_broyden_and_anderson(pa, dtype, kind):
    A, b, lam, Lf = lasso_small(dtype)
    x0 = np.zeros(5, dtype)
    tag = lambda: pa.Broyden() if kind == "broyden" else pa.AndersonAcceleration(5)
    z, it = pa.DRLS(tol=10 * rv.LASSO_SMALL_TOL, directions=tag())(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam), Lf=Lf)
    assert np.max(np.abs(z - rv.LASSO_SMALL_XSTAR.astype(dtype))) <= 10 * rv.LASSO_SMALL_TOL
    assert it < rv.LASSO_SMALL_BOUNDS_EXT["drls_" + kind]
    _, ito = ox.drls(tol=10 * rv.LASSO_SMALL_TOL, directions=kind, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=Lf)
    assert it == ito if dtype == np.float64 else abs(it - ito) <= 3
    # the same operators plug into PANOC / ZeroFPR (QuasiNewtonStyle, panoc.jl:114-126)
    for solver in (pa.PANOC, pa.ZeroFPR):
        x, it = solver(tol=rv.LASSO_SMALL_TOL, directions=tag())(x0=x0, f=pa.SquaredDistance(b), A=A, g=pa.NormL1(lam), Lf=Lf)
        assert np.max(np.abs(x - rv.LASSO_SMALL_XSTAR.astype(dtype))) <= rv.LASSO_SMALL_TOL and it < 40


# ------------------------------------------------------------------------------------------------
# linear programs (test/problems/test_linear_programs.jl): Linear, IndNonnegative, IndPoint, IndAffine, SlicedSeparableSum
# ------------------------------------------------------------------------------------------------


def lp_problem(dtype):
    A = np.asfortranarray(rv.LP_A.astype(dtype))
    b = A @ rv.LP_XSTAR.astype(dtype)
    c = A.T @ rv.LP_YSTAR.astype(dtype) + rv.LP_SSTAR.astype(dtype)
    return A, b, c, 100 * np.finfo(dtype).eps


def assert_lp_solution(c, A, b, x, y, tol):
    assert -min(0.0, float(x.min())) <= tol
    assert np.linalg.norm(A @ x - b) <= tol
    assert max(0.0, float((-A.T @ y - c).max())) <= tol
    assert abs(float(np.dot(c + A.T @ y, x))) <= tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_programs_on_device(pa, dtype):
    A, b, c, tol = lp_problem(dtype)
    n, m = 10, 8
    x0 = np.zeros(n, dtype)
    for solver, osolver in ((pa.AFBA, ox.afba), (pa.VuCondat, ox.vu_condat)):
        (x, y), it = solver(tol=tol, maxit=rv.LP_MAXIT)(x0=x0, y0=np.zeros(m, dtype), f=pa.Linear(c), g=pa.IndNonnegative(),
                                                        h=pa.IndPoint(b), L=A, beta_f=0)
        assert x.dtype == dtype and y.dtype == dtype and it <= rv.LP_MAXIT and np.all(x0 == 0)
        assert_lp_solution(c, A, b, x, y, 1000 * tol)
        _, ito = osolver(tol=tol, maxit=rv.LP_MAXIT, x0=x0, y0=np.zeros(m, dtype), f=ox.Linear(c), g=ox.IndNonnegative(),
                         h=ox.IndPoint(b), L=A, beta_f=0)
        assert abs(it - ito) <= max(3, ito // 5)  # the rule fires at 100 eps: rounding-floor territory in both precisions
    xf, it = pa.DavisYin(gamma=dtype(1), tol=tol, maxit=rv.LP_MAXIT)(x0=x0, f=pa.Linear(c), g=pa.IndNonnegative(),
                                                                    h=pa.IndAffine(A, b))
    assert xf.dtype == dtype and it <= rv.LP_MAXIT
    assert np.linalg.norm(xf - rv.LP_XSTAR.astype(dtype)) <= 100 * tol
    if dtype == np.float32:  # ChambollePock needs ~20k iterations in Float32, ~80k in Float64 (oracle-pinned on the CPU)
        h = pa.SlicedSeparableSum((pa.IndPoint(b), pa.IndNonnegative()), ((0, m), (m, m + n)))
        (x, y), it = pa.ChambollePock(tol=tol, maxit=rv.LP_MAXIT)(x0=x0, y0=np.zeros(m + n, dtype), g=pa.Linear(c), h=h,
                                                                  L=np.vstack([A, np.eye(n, dtype=dtype)]))
        assert it <= rv.LP_MAXIT
        assert_lp_solution(c, A, b, x, y[:m], 1000 * tol)


def test_verbose_display_second_group(pa, capsys):
    """test/problems/test_verbose.jl: the display path of every solver runs and does not change the answer"""
    dtype = np.float64
    A, b, lam, Lf = lasso_small(dtype)
    x0 = np.zeros(5, dtype)
    xs = rv.LASSO_SMALL_XSTAR
    f, g = pa.LeastSquares(A, b), pa.NormL1(lam)
    y, _ = pa.SFISTA(tol=1e-4, verbose=True, freq=5)(x0=x0, f=f, g=g, Lf=Lf)
    assert np.max(np.abs(y - xs)) <= 1e-3
    z, _ = pa.DRLS(tol=1e-4, verbose=True, freq=1)(x0=x0, f=f, g=g, Lf=Lf)
    assert np.max(np.abs(z - xs)) <= 1e-3
    z, _ = pa.LiLin(tol=1e-4, verbose=True, freq=20)(x0=x0, f=f, g=g, Lf=Lf)
    assert np.max(np.abs(z - xs)) <= 1e-3
    x, _ = pa.DavisYin(tol=1e-6, verbose=True, freq=20)(x0=x0, f=f, g=pa.NormL1(1.0), h=pa.SqrNormL2(1.0), Lf=Lf)
    assert np.max(np.abs(x - rv.ELASTICNET_XSTAR)) <= 1e-3
    (x, _), _ = pa.AFBA(tol=1e-6, verbose=True, freq=20)(x0=x0, y0=np.zeros(5), f=f, g=g, beta_f=Lf)
    assert np.max(np.abs(x - xs)) <= 1e-4
    out = capsys.readouterr().out
    assert out.count("\n") >= 10 and "|" in out


# ------------------------------------------------------------------------------------------------
# hipGraph replay of launch-bound iteration bodies (pg_ctx_capture_begin / _end, pg_graph_launch)
# ------------------------------------------------------------------------------------------------


@pytest.fixture()
def stream_ctx(pa):
    """a context on its own stream (the null stream cannot be captured) as the process default for one test"""
    ctx = pa.Context.on_new_stream()
    prev = pa.set_default_context(ctx)
    yield ctx
    ctx.sync()
    if prev is not None:
        pa.set_default_context(prev)
    else:
        from proximalalgorithms.jl_amd import device

        device._default_ctx.pop(ctx.device, None)


@pytest.mark.parametrize("dtype", DTYPES)
def test_graph_replay_matches_plain_stepping(pa, stream_ctx, dtype):
    from proximalalgorithms.jl_amd.algorithm import graph_iterate

    rng = np.random.default_rng(17)
    m, n = 60, 90
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    b = rng.standard_normal(m).astype(dtype)
    R = np.dtype(dtype).type
    lam = R(0.1) * R(np.max(np.abs(A.T @ b)))
    Lf = R(np.linalg.norm(A, 2) ** 2)
    x0, y0 = np.zeros(n, dtype), np.zeros(m, dtype)
    makers = {
        "afba": lambda: pa.AFBAIteration(x0=x0, y0=y0, f=pa.SqrNormL2(R(0.5)), beta_f=0.5, g=pa.NormL1(lam),
                                         h=pa.SquaredDistance(b), L=A),
        "afba_ls": lambda: pa.AFBAIteration(x0=x0, y0=np.zeros(n, dtype), f=pa.LeastSquares(A, b), g=pa.NormL1(lam), beta_f=Lf),
        "chambolle_pock_single_sweep": lambda: pa.ChambollePockIteration(x0=x0, y0=y0, g=pa.NormL1(lam), h=pa.SquaredDistance(b), L=A),
        "davis_yin": lambda: pa.DavisYinIteration(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam), h=pa.IndBox(-0.5, 0.5), Lf=Lf,
                                                  single_sweep=False),  # the plain body (the sweep form swaps buffers)
        "dr": lambda: pa.DouglasRachfordIteration(x0=x0, f=pa.LeastSquares(A, b), g=pa.NormL1(lam), gamma=R(1) / Lf),
    }
    for name, make in makers.items():
        plain_it, graph_it = make(), make()
        assert plain_it.x0.ctx is stream_ctx and graph_it.graph_safe
        plain, replay = iter(plain_it), graph_iterate(graph_it)
        for k in range(12):
            sp, sg = next(plain), next(replay)
            key = "x" if name != "davis_yin" else "z"
            assert np.array_equal(getattr(sp, key).numpy(), getattr(sg, key).numpy()), (name, k)
        assert graph_it.graph is not None, name  # really replayed, not the fallback
    # the drivers take graph=True and give the same answer and iteration count
    for solver, kw in ((pa.AFBA, dict(y0=np.zeros(n, dtype), f=pa.LeastSquares(A, b), g=pa.NormL1(lam), beta_f=Lf)),
                       (pa.DavisYin, dict(f=pa.LeastSquares(A, b), g=pa.NormL1(lam), h=pa.IndBox(-0.5, 0.5), Lf=Lf)),
                       (pa.DouglasRachford, dict(f=pa.LeastSquares(A, b), g=pa.NormL1(lam), gamma=R(1) / Lf))):
        tol = R(1e-4 if dtype == np.float32 else 1e-8)
        a, ita = solver(tol=tol, maxit=3000)(x0=x0, **kw)
        g, itg = solver(tol=tol, maxit=3000, graph=True)(x0=x0, **kw)
        a, g = (a[0], g[0]) if isinstance(a, tuple) else (a, g)
        assert ita == itg and np.array_equal(a, g), solver.__name__


def test_graph_replay_of_a_team_sweep(pa, stream_ctx):
    """a recorded long-column sweep replays with its kernel arguments baked in, so the exchange ring's
    launch epoch cannot advance between replays; with few columns (every step within the ring's 8 slots) a replay would
    meet its own granules of the previous replay.  The recorded body zeroes the ring itself (reserved epoch): Chambolle-Pock
    with a 40000 x 48 device matrix L (teams of three workgroups, one step per team) replayed 12 times against plain
    stepping, bit for bit, and a plain step after the replays (its epoch must not collide with the recorded one)."""
    from proximalalgorithms.jl_amd.algorithm import graph_iterate

    dtype = np.float32
    rng = np.random.default_rng(23)
    m, n = 40000, 48
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    b = rng.standard_normal(m).astype(dtype)
    lam = dtype(0.1) * dtype(np.max(np.abs(A.T @ b)))
    L = pa.HIPMatrix.from_numpy(A, stream_ctx)
    make = lambda: pa.ChambollePockIteration(x0=np.zeros(n, dtype), y0=np.zeros(m, dtype), g=pa.NormL1(lam),
                                             h=pa.SquaredDistance(b), L=L)
    plain_it, graph_it = make(), make()
    plain, replay = iter(plain_it), graph_iterate(graph_it)
    for k in range(12):
        sp, sg = next(plain), next(replay)
        assert np.array_equal(sp.x.numpy(), sg.x.numpy()) and np.array_equal(sp.y.numpy(), sg.y.numpy()), k
    assert graph_it.graph is not None and graph_it.single_sweep and plain_it.single_sweep
    assert np.any(sp.x.numpy() != 0)
    # uncaptured launches on the same matrix after the replays
    more = iter(make())
    ref = iter(make())
    for k in range(3):
        s1, s2 = next(more), next(ref)
        assert np.array_equal(s1.x.numpy(), s2.x.numpy()), k


def test_graph_falls_back_on_default_stream_and_refuses_allocation(pa, stream_ctx):
    from proximalalgorithms.jl_amd import device
    from proximalalgorithms.jl_amd.algorithm import graph_iterate

    A, b, lam, Lf = lasso_small(np.float64)
    # allocation inside a capture is refused (its address would be baked into the graph)
    stream_ctx.capture_begin()
    with pytest.raises(pa.ProxGradError):
        pa.HIPVector.empty(5, np.float64)
    assert stream_ctx.capture_end(abort=True) is None
    v = pa.HIPVector.zeros(5, np.float64)  # the context still works after the aborted capture
    assert np.all(v.numpy() == 0)
    # a context on the default stream cannot capture: graph=True silently steps without a graph
    dflt = pa.Context()
    prev = pa.set_default_context(dflt)
    try:
        it = pa.AFBAIteration(x0=np.zeros(5), y0=np.zeros(5), f=pa.LeastSquares(A, b), g=pa.NormL1(lam), beta_f=Lf)
        steps = graph_iterate(it)
        for _ in range(5):
            s = next(steps)
        assert it.graph is None and np.all(np.isfinite(s.x.numpy()))
        (x, _), k = pa.AFBA(theta=1, mu=1, tol=1e-6, graph=True)(x0=np.zeros(5), y0=np.zeros(5), f=pa.LeastSquares(A, b),
                                                                  g=pa.NormL1(lam), beta_f=Lf)
        assert np.max(np.abs(x - rv.LASSO_SMALL_XSTAR)) <= 1e-4
    finally:
        pa.set_default_context(prev)
    assert device.get_context() is stream_ctx


@pytest.mark.parametrize("dtype", DTYPES)
def test_afba_with_infimal_convolution_term(pa, dtype):
    """the full template f + g + (h [] l)(L x) with a strongly convex l (primal_dual.jl:187: gradient of l*):
    h = lam ||.||_1, l = beta/2 ||.||^2  ->  h [] l is the Huber function; checked against the restatement and
    against the stationarity of the smoothed problem."""
    rng = np.random.default_rng(31)
    m, n = 40, 25
    L = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    c = rng.standard_normal(n).astype(dtype)
    R = np.dtype(dtype).type
    lam, beta = R(0.3), R(2.0)
    x0, y0 = np.zeros(n, dtype), np.zeros(m, dtype)
    tol = R(1e-4 if dtype == np.float32 else 1e-9)
    kw = dict(x0=x0, y0=y0, beta_f=1, beta_l=float(1 / beta))
    for theta, mu in ((2, 0), (1, 1), (0, 1), (0, 0.5), (1, 0), (0, 0)):
        (x, y), it = pa.AFBA(theta=theta, mu=mu, tol=tol)(f=pa.SquaredDistance(c), h=pa.NormL1(lam), l=pa.SqrNormL2(beta), L=L, **kw)
        (xo, yo), ito = ox.afba(theta=theta, mu=mu, tol=tol, f=ox.SqrDistance(c), h=o.NormL1(lam), l=ox.SqrNormL2(beta), L=L, **kw)
        assert abs(it - ito) <= (max(3, ito // 10) if dtype == np.float32 else 0), (theta, mu)
        assert close(x, xo, dtype, 5) and close(y, yo, dtype, 5), (theta, mu)
        # stationarity: x - c + L' huber'(L x) = 0 with huber'(z) = clip(beta z, -lam, lam)
        x64 = x.astype(np.float64)
        z = L.astype(np.float64) @ x64
        grad = x64 - c + L.astype(np.float64).T @ np.clip(float(beta) * z, -float(lam), float(lam))
        assert np.max(np.abs(grad)) <= (2e-3 if dtype == np.float32 else 1e-7), (theta, mu)


def test_fuzz_second_group_against_oracle(pa):
    """tests/tools/fuzz_second_group.py: random shapes / operator pairs / iterations, iterate by iterate (Float64);
    2000 cases were run clean in round 1"""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_second_group", os.path.join(root, "tests", "tools", "fuzz_second_group.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert not mod.run(150, first_seed=5000)


@pytest.mark.parametrize("dtype", DTYPES)
def test_vu_condat_single_sweep_one_read_of_L_per_iteration(pa, dtype):
    """theta = 2: the sweep gives L'y, the primal prox and L xbar; L (2 xbar - x) follows by linearity -> ONE read of L
    per iteration (four products in the reference's statement order), same iterates."""
    rng = np.random.default_rng(41)
    m, n = 90, 260
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    b = rng.standard_normal(m).astype(dtype)
    R = np.dtype(dtype).type
    lam = R(0.1) * R(np.max(np.abs(A.T @ b)))
    x0, y0 = (0.1 * rng.standard_normal(n)).astype(dtype), (0.1 * rng.standard_normal(m)).astype(dtype)
    Ad = pa.HIPMatrix.from_numpy(A)
    cases = {
        "chambolle_pock": (dict(g=pa.NormL1(lam), h=pa.SquaredDistance(b)), dict(g=o.NormL1(lam), h=ox.SqrDistance(b))),
        "vu_condat_f": (dict(f=pa.SqrNormL2(R(0.5)), beta_f=0.5, g=pa.IndBox(-0.3, 0.4), h=pa.SquaredDistance(b)),
                        dict(f=ox.SqrNormL2(R(0.5)), beta_f=0.5, g=o.IndBox(-0.3, 0.4), h=ox.SqrDistance(b))),
        "vu_condat_l": (dict(f=pa.SqrNormL2(R(0.5)), beta_f=0.5, g=pa.NormL1(lam), h=pa.NormL1(R(0.2)), l=pa.SqrNormL2(R(2.0)), beta_l=0.5),
                        dict(f=ox.SqrNormL2(R(0.5)), beta_f=0.5, g=o.NormL1(lam), h=o.NormL1(R(0.2)), l=ox.SqrNormL2(R(2.0)), beta_l=0.5)),
    }
    K = 30
    for name, (kd, ko) in cases.items():
        one = pa.AFBAIteration(x0=x0, y0=y0, L=Ad, theta=2, **kd)
        four = pa.AFBAIteration(x0=x0, y0=y0, L=Ad, theta=2, single_sweep=False, **kd)
        ora = ox.AFBAIteration(x0=x0, y0=y0, L=A, theta=2, **ko)
        assert one.single_sweep and not four.single_sweep
        for k, (s1, s4, so) in enumerate(zip(one, four, ora)):
            if k >= K:
                break
            for fld in ("x", "y", "xbar", "ybar"):
                ref = getattr(so, fld)
                tol = (2e-4 if dtype == np.float32 else 1e-10) * max(1.0, float(np.max(np.abs(ref))))
                assert np.max(np.abs(getattr(s1, fld).numpy() - ref)) <= tol, (name, k, fld)
                assert np.max(np.abs(getattr(s1, fld).numpy() - getattr(s4, fld).numpy())) <= tol, (name, k, fld)
        assert one.counters["L_passes"] == K + 2, name  # one sweep per iteration + L x0 (the (K+1)-th body ran in zip)
        assert four.counters["L_passes"] == 2 * (K + 1), name  # L'y and L(2 xbar - x); the zero-factor products are skipped
    # the driver, with and without the hipGraph, gives the same answer
    tol = R(1e-4 if dtype == np.float32 else 1e-8)
    (xa, ya), ita = pa.ChambollePock(tol=tol)(x0=x0, y0=y0, g=pa.NormL1(lam), h=pa.SquaredDistance(b), L=Ad)
    (xo, yo), ito = ox.chambolle_pock(tol=tol, x0=x0, y0=y0, g=o.NormL1(lam), h=ox.SqrDistance(b), L=A)
    assert abs(ita - ito) <= (max(3, ito // 10) if dtype == np.float32 else 1) and close(xa, xo, dtype, 5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_sfista_single_sweep_pass_counts_and_refresh(pa, dtype):
    """f = LeastSquares on a device matrix: 1 read of A per iteration + 1 for the default stop rule (4 in the reference's
    statement order); A x rides its linear recurrence and is recomputed every 64 iterations -- 200 iterations stay on the
    plain path's iterates."""
    rng = np.random.default_rng(8)
    m, n = 150, 400
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    b = rng.standard_normal(m).astype(dtype)
    R = np.dtype(dtype).type
    lam = R(0.05) * R(np.max(np.abs(A.T @ b)))
    Lf = R(np.linalg.norm(A, 2) ** 2)
    x0 = np.zeros(n, dtype)
    Ad = pa.HIPMatrix.from_numpy(A)
    one = pa.SFISTAIteration(x0=x0, f=pa.LeastSquares(Ad, b), g=pa.NormL1(lam), Lf=Lf, mf=R(0.01))
    two = pa.SFISTAIteration(x0=x0, f=pa.LeastSquares(Ad, b), g=pa.NormL1(lam), Lf=Lf, mf=R(0.01), single_sweep=False)
    K = 200
    for k, (s1, s2) in enumerate(zip(one, two)):
        if k >= K:
            break
        tol = (5e-4 if dtype == np.float32 else 1e-9) * max(1.0, float(np.max(np.abs(s2.y.numpy()))))
        assert np.max(np.abs(s1.y.numpy() - s2.y.numpy())) <= tol, k
    assert one.counters["a_passes"] == 1 + (K + 1) + (K + 1) // 64
    from proximalalgorithms.jl_amd.sfista import check_sc

    r1, _ = check_sc(s1, one, 1e-3)
    r2, _ = check_sc(s2, two, 1e-3)
    assert float(r1) == pytest.approx(float(r2), rel=1e-2 if dtype == np.float32 else 1e-6, abs=1e-6)
    assert one.counters["a_passes"] == 1 + (K + 1) + (K + 1) // 64 + 1


def test_custom_operators_plug_into_the_second_group(pa, stream_ctx):
    """docs/src/guide/custom_objectives.jl: any object with prox_ / value_and_gradient methods works; operators without
    the `want_value` / `out` keywords and operators that allocate still run (value computed and dropped, gradient copied,
    hipGraph capture refused -> plain stepping)."""
    dtype = np.float64
    A, b, lam, Lf = lasso_small(dtype)

    class MyL1:  # no want_value keyword, allocates a temporary on every call
        def __init__(self, lam):
            self.inner = pa.NormL1(lam)

        def prox_(self, y, x, gamma):
            tmp = x.similar().copy_from(x)
            return self.inner.prox_(y, tmp, gamma)

    class MyLeastSquares:  # no `out` keyword
        def __init__(self):
            self.inner = pa.LeastSquares(A, b)

        def value_and_gradient(self, x):
            return self.inner.value_and_gradient(x)

        def __call__(self, x):  # li_lin.jl:103 evaluates iter.f(z)
            return self.inner(x)

    x0 = np.zeros(5, dtype)
    xs = rv.LASSO_SMALL_XSTAR
    (x, _), it = pa.AFBA(theta=1, mu=1, tol=1e-6, graph=True)(x0=x0, y0=np.zeros(5), f=MyLeastSquares(), g=MyL1(lam), beta_f=Lf)
    (xr, _), itr = pa.AFBA(theta=1, mu=1, tol=1e-6)(x0=x0, y0=np.zeros(5), f=pa.LeastSquares(A, b), g=pa.NormL1(lam), beta_f=Lf)
    assert it == itr and np.max(np.abs(x - xr)) <= 1e-12 and np.max(np.abs(x - xs)) <= 1e-4
    y, it = pa.SFISTA(tol=1e-4)(x0=x0, f=MyLeastSquares(), g=MyL1(lam), Lf=Lf)
    assert np.max(np.abs(y - xs)) <= 1e-3
    x, it = pa.DavisYin(tol=1e-7, graph=True)(x0=x0, f=MyLeastSquares(), g=MyL1(lam), h=pa.Zero(), Lf=Lf)
    assert np.max(np.abs(x - xs)) <= 1e-4
    z, it = pa.LiLin(tol=1e-5)(x0=x0, f=MyLeastSquares(), g=pa.NormL1(lam), Lf=Lf)
    assert np.max(np.abs(z - xs)) <= 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_afba_default_two_reads_of_L_per_iteration(pa, dtype):
    """theta = 1, mu = 1 with a device matrix L: the sweep (L'y, prox, L xbar) plus the primal correction -- two reads of L per
    iteration instead of three, same iterates as the plain statement order and the restatement"""
    rng = np.random.default_rng(43)
    m, n = 70, 210
    A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
    b = rng.standard_normal(m).astype(dtype)
    R = np.dtype(dtype).type
    lam = R(0.1) * R(np.max(np.abs(A.T @ b)))
    x0, y0 = (0.1 * rng.standard_normal(n)).astype(dtype), (0.1 * rng.standard_normal(m)).astype(dtype)
    Ad = pa.HIPMatrix.from_numpy(A)
    kd = dict(f=pa.SqrNormL2(R(0.3)), beta_f=0.3, g=pa.NormL1(lam), h=pa.SquaredDistance(b), lam=R(0.9), gamma=(R(0.2), R(0.3)))
    ko = dict(f=ox.SqrNormL2(R(0.3)), beta_f=0.3, g=o.NormL1(lam), h=ox.SqrDistance(b), lam=R(0.9), gamma=(R(0.2), R(0.3)))
    two = pa.AFBAIteration(x0=x0, y0=y0, L=Ad, **kd)
    three = pa.AFBAIteration(x0=x0, y0=y0, L=Ad, single_sweep=False, **kd)
    ora = ox.AFBAIteration(x0=x0, y0=y0, L=A, **ko)
    assert two.single_sweep and not three.single_sweep
    K = 30
    for k, (s2, s3, so) in enumerate(zip(two, three, ora)):
        if k >= K:
            break
        for fld in ("x", "y", "xbar", "ybar"):
            ref = getattr(so, fld)
            tol = (2e-4 if dtype == np.float32 else 1e-10) * max(1.0, float(np.max(np.abs(ref))))
            assert np.max(np.abs(getattr(s2, fld).numpy() - ref)) <= tol, (k, fld)
            assert np.max(np.abs(getattr(s2, fld).numpy() - getattr(s3, fld).numpy())) <= tol, (k, fld)
    assert two.counters["L_passes"] == 2 * (K + 1) and three.counters["L_passes"] == 3 * (K + 1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_davis_yin_single_sweep_one_read_of_A_per_iteration(pa, dtype):
    """f = LeastSquares / Composed on a device matrix, g and h in-kernel prox kinds: the whole Davis-Yin iteration (and the
    next prox_g point with its image) in ONE read of A (pg_mat_fused_dys); same iterates as the plain statement order
    (two reads) and the restatement."""
    rng = np.random.default_rng(47)
    R = np.dtype(dtype).type
    for (m, n) in ((60, 200), (300, 90), (1100, 40)):
        A = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype) / dtype(np.sqrt(m)))
        b = rng.standard_normal(m).astype(dtype)
        lam = R(0.1) * R(np.max(np.abs(A.T @ b)))
        Lf = R(np.linalg.norm(A, 2) ** 2)
        x0 = (0.1 * rng.standard_normal(n)).astype(dtype)
        Ad = pa.HIPMatrix.from_numpy(A)
        cases = [
            (dict(f=pa.LeastSquares(Ad, b), g=pa.NormL1(lam), h=pa.SqrNormL2(R(0.7))), dict(f=o.LeastSquares(A, b), g=o.NormL1(lam), h=ox.SqrNormL2(R(0.7)))),
            (dict(f=pa.Composed(pa.LogisticLoss(b), Ad), g=pa.IndBox(-0.4, 0.3), h=pa.NormL1(R(0.01))),
             dict(f=o.Composed(o.LogisticLoss(b), A), g=o.IndBox(-0.4, 0.3), h=o.NormL1(R(0.01)))),
            (dict(f=pa.LeastSquares(Ad, b, lam=0.5), g=pa.Zero(), h=pa.IndBox(-0.2, 0.2)), dict(f=o.LeastSquares(A, b, 0.5), g=o.Zero(), h=o.IndBox(-0.2, 0.2))),
        ]
        K = 25
        for kd, ko in cases:
            one = pa.DavisYinIteration(x0=x0, Lf=Lf, lam=R(1.2), **kd)
            two = pa.DavisYinIteration(x0=x0, Lf=Lf, lam=R(1.2), single_sweep=False, **kd)
            ora = ox.DavisYinIteration(x0=x0, Lf=Lf, lam=R(1.2), **ko)
            assert one._sweep is not None and two._sweep is None and not one.graph_safe and two.graph_safe
            for k, (s1, s2, so) in enumerate(zip(one, two, ora)):
                if k >= K:
                    break
                for fld in ("z", "xg", "xh", "res", "z_half", "grad_f_xg"):
                    ref = getattr(so, fld)
                    tol = (3e-4 if dtype == np.float32 else 1e-10) * max(1.0, float(np.max(np.abs(ref))))
                    assert np.max(np.abs(getattr(s1, fld).numpy() - ref)) <= tol, (m, n, k, fld)
                    assert np.max(np.abs(getattr(s1, fld).numpy() - getattr(s2, fld).numpy())) <= tol, (m, n, k, fld)
                assert float(s1.res_inf) == pytest.approx(float(np.max(np.abs(so.res))), rel=1e-3, abs=1e-6)
            assert one.counters["a_passes"] == 1 + (K + 1)
