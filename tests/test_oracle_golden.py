"""Pin the CPU oracle (oracle/proxgrad_oracle.py) against the reference's own known answers.

Mirrors: test/problems/test_lasso_small.jl, test_lasso_small_strongly_convex.jl,
test/accel/test_lbfgs.jl, test/accel/test_nesterov.jl, test/utilities/test_fb_tools.jl,
test/problems/test_equivalence.jl:51-84, test/problems/test_nonconvex_qp.jl:33-34 and the
stored optima of benchmark/data/lasso_*.jld2.
"""
import itertools
import os

import numpy as np
import pytest

import reference_vectors as rv
from oracle import proxgrad_oracle as o

GOLDEN = os.path.dirname(os.path.abspath(rv.__file__))


def lasso_small(dtype):
    A = np.asfortranarray(rv.LASSO_SMALL_A.astype(dtype))
    b = rv.LASSO_SMALL_B.astype(dtype)
    R = np.dtype(dtype).type
    lam = R(0.1) * R(np.max(np.abs(A.T @ b)))
    Lf = R(np.linalg.norm(A, 2) ** 2)
    return A, b, lam, Lf


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
class TestLassoSmall:
    """test/problems/test_lasso_small.jl:46-135"""

    def check(self, dtype, x, it, bound, x0):
        assert x.dtype == dtype
        assert np.max(np.abs(x - rv.LASSO_SMALL_XSTAR.astype(dtype))) <= rv.LASSO_SMALL_TOL
        assert it < bound
        assert np.all(x0 == 0)

    def test_fb_fixed(self, dtype):
        A, b, lam, Lf = lasso_small(dtype)
        x0 = np.zeros(5, dtype)
        x, it = o.forward_backward(tol=rv.LASSO_SMALL_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=Lf)
        self.check(dtype, x, it, rv.LASSO_SMALL_BOUNDS["fb_fixed"], x0)

    def test_fb_adaptive(self, dtype):
        A, b, lam, Lf = lasso_small(dtype)
        x0 = np.zeros(5, dtype)
        x, it = o.forward_backward(tol=rv.LASSO_SMALL_TOL, adaptive=True, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam))
        self.check(dtype, x, it, rv.LASSO_SMALL_BOUNDS["fb_adaptive"], x0)

    def test_fb_adaptive_regret(self, dtype):
        A, b, lam, Lf = lasso_small(dtype)
        x0 = np.zeros(5, dtype)
        x, it = o.forward_backward(tol=rv.LASSO_SMALL_TOL, adaptive=True, increase_gamma=1.01, x0=x0,
                                   f=o.LeastSquares(A, b), g=o.NormL1(lam))
        self.check(dtype, x, it, rv.LASSO_SMALL_BOUNDS["fb_adaptive_regret"], x0)

    def test_ffb_fixed(self, dtype):
        A, b, lam, Lf = lasso_small(dtype)
        x0 = np.zeros(5, dtype)
        x, it = o.fast_forward_backward(tol=rv.LASSO_SMALL_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=Lf)
        self.check(dtype, x, it, rv.LASSO_SMALL_BOUNDS["ffb_fixed"], x0)

    def test_ffb_adaptive(self, dtype):
        A, b, lam, Lf = lasso_small(dtype)
        x0 = np.zeros(5, dtype)
        x, it = o.fast_forward_backward(tol=rv.LASSO_SMALL_TOL, adaptive=True, x0=x0, f=o.LeastSquares(A, b),
                                        g=o.NormL1(lam))
        self.check(dtype, x, it, rv.LASSO_SMALL_BOUNDS["ffb_adaptive"], x0)

    def test_ffb_adaptive_regret(self, dtype):
        A, b, lam, Lf = lasso_small(dtype)
        x0 = np.zeros(5, dtype)
        x, it = o.fast_forward_backward(tol=rv.LASSO_SMALL_TOL, adaptive=True, increase_gamma=1.01, x0=x0,
                                        f=o.LeastSquares(A, b), g=o.NormL1(lam))
        self.check(dtype, x, it, rv.LASSO_SMALL_BOUNDS["ffb_adaptive_regret"], x0)

    def test_ffb_custom_sequence(self, dtype):
        A, b, lam, Lf = lasso_small(dtype)
        x0 = np.zeros(5, dtype)
        x, it = o.fast_forward_backward(tol=rv.LASSO_SMALL_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=Lf,
                                        extrapolation_sequence=o.fixed_nesterov_sequence(dtype))
        self.check(dtype, x, it, rv.LASSO_SMALL_BOUNDS["ffb_fixed_custom_seq"], x0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
class TestLassoStronglyConvex:
    """test/problems/test_lasso_small_strongly_convex.jl:65-144"""

    def setup_problem(self, dtype):
        A, b, lam, x0 = rv.strongly_convex_problem(dtype)
        return A, b, lam, x0, x0.copy()

    def check(self, dtype, y, it, bound, x0, x0_backup):
        assert y.dtype == dtype
        assert np.max(np.abs(y - rv.SC_XSTAR.astype(dtype))) <= rv.SC_TOL
        assert it < bound
        assert np.array_equal(x0, x0_backup)

    def test_fb(self, dtype):
        A, b, lam, x0, bk = self.setup_problem(dtype)
        y, it = o.forward_backward(tol=rv.SC_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=dtype(rv.SC_LF))
        self.check(dtype, y, it, rv.SC_BOUNDS["fb_fixed"], x0, bk)

    def test_fb_adaptive(self, dtype):
        A, b, lam, x0, bk = self.setup_problem(dtype)
        y, it = o.forward_backward(tol=rv.SC_TOL, adaptive=True, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam))
        self.check(dtype, y, it, rv.SC_BOUNDS["fb_adaptive"], x0, bk)

    def test_fb_adaptive_regret(self, dtype):
        A, b, lam, x0, bk = self.setup_problem(dtype)
        y, it = o.forward_backward(tol=rv.SC_TOL, adaptive=True, increase_gamma=1.01, x0=x0, f=o.LeastSquares(A, b),
                                   g=o.NormL1(lam))
        self.check(dtype, y, it, rv.SC_BOUNDS["fb_adaptive_regret"], x0, bk)

    def test_ffb_mf(self, dtype):
        A, b, lam, x0, bk = self.setup_problem(dtype)
        y, it = o.fast_forward_backward(tol=rv.SC_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam),
                                        Lf=dtype(rv.SC_LF), mf=dtype(rv.SC_MF))
        self.check(dtype, y, it, rv.SC_BOUNDS["ffb_fixed_mf"], x0, bk)

    def test_ffb_adaptive(self, dtype):
        A, b, lam, x0, bk = self.setup_problem(dtype)
        y, it = o.fast_forward_backward(tol=rv.SC_TOL, adaptive=True, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam))
        self.check(dtype, y, it, rv.SC_BOUNDS["ffb_adaptive"], x0, bk)

    def test_ffb_adaptive_regret(self, dtype):
        A, b, lam, x0, bk = self.setup_problem(dtype)
        y, it = o.fast_forward_backward(tol=rv.SC_TOL, adaptive=True, increase_gamma=1.01, x0=x0,
                                        f=o.LeastSquares(A, b), g=o.NormL1(lam))
        self.check(dtype, y, it, rv.SC_BOUNDS["ffb_adaptive_regret"], x0, bk)

    def test_ffb_constant_sequence(self, dtype):
        A, b, lam, x0, bk = self.setup_problem(dtype)
        mf, Lf = dtype(rv.SC_MF), dtype(rv.SC_LF)
        y, it = o.fast_forward_backward(tol=rv.SC_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam),
                                        gamma=dtype(1) / Lf, mf=mf,
                                        extrapolation_sequence=o.constant_nesterov_sequence(mf, dtype(1) / Lf))
        self.check(dtype, y, it, rv.SC_BOUNDS["ffb_constant_seq"], x0, bk)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lbfgs_golden_directions(dtype):
    """test/accel/test_lbfgs.jl:103-133"""
    Q, q = rv.LBFGS_Q.astype(dtype), rv.LBFGS_q.astype(dtype)
    xs = rv.LBFGS_XS.astype(dtype)
    H = o.LBFGSOperator(rv.LBFGS_MEM, np.zeros(10, dtype))
    x = xs[0]
    grad = Q @ x + q
    d = -(H * grad)
    rtol = 1e-4 if dtype == np.float32 else 1e-8  # Julia isapprox default rtol = sqrt(eps)
    np.testing.assert_allclose(d, rv.LBFGS_DIRS_REF[0], rtol=rtol, atol=rtol * np.linalg.norm(rv.LBFGS_DIRS_REF[0]))
    for i in range(1, 5):
        x_prev, grad_prev = x, grad
        x = xs[i]
        grad = Q @ x + q
        H.update(x - x_prev, grad - grad_prev)
        H.mul(d, -grad)
        ref = rv.LBFGS_DIRS_REF[i]
        assert np.linalg.norm(d - ref) <= np.sqrt(np.finfo(dtype).eps) * max(np.linalg.norm(d), np.linalg.norm(ref))
    H.reset()
    assert np.array_equal(H * x, x)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("seq", ["simple", "fixed"])
def test_nesterov_beck_teboulle_bound(dtype, seq):
    """test/accel/test_nesterov.jl:12-61"""
    R = np.dtype(dtype).type
    Hm, l = rv.NESTEROV_H.astype(dtype), rv.NESTEROV_l.astype(dtype)
    f = lambda x: np.dot(x, Hm @ x) / 2 + np.dot(x, l)
    x_star = -np.linalg.solve(Hm, l)
    f_star = f(x_star)
    gamma = R(1) / R(np.linalg.norm(Hm, 2))
    x = np.zeros(5, dtype)
    y = x
    err0 = np.linalg.norm(x_star - x)
    gen = o.simple_nesterov_sequence(dtype) if seq == "simple" else o.fixed_nesterov_sequence(dtype)
    for it, coeff in enumerate(itertools.islice(gen, 100), start=1):
        assert coeff.dtype == dtype
        if it == 1:
            assert coeff == 0
        g = Hm @ y + l
        x_prev = x
        x = y - gamma * g
        y = x + coeff * (x - x_prev)
        assert f(x) - f_star <= 2 / (gamma * (it + 1) ** 2) * err0**2


def test_nesterov_adaptive_identities():
    """test/accel/test_nesterov.jl:63-81"""
    R = np.float64
    g = R(rv.NESTEROV_FIXED_GAMMA)
    ad = o.AdaptiveNesterovSequence(R(0))
    for el in itertools.islice(o.fixed_nesterov_sequence(R), 20):
        assert np.isclose(el, ad.next(g), rtol=1e-8)
    m = R(1)
    ad = o.AdaptiveNesterovSequence(m)
    for _ in range(20):
        assert np.isclose((1 - np.sqrt(m * g)) / (1 + np.sqrt(m * g)), ad.next(g), rtol=1e-8)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fb_tools_properties(dtype):
    """test/utilities/test_fb_tools.jl:7-48"""
    rng = np.random.default_rng(0)
    sv = np.array([0.01, 1.0, 1.0, 1.0, 100.0], dtype)
    U, _ = np.linalg.qr(rng.standard_normal((5, 5)).astype(dtype))
    Q = ((U * sv[None, :]) @ U.T).astype(dtype)
    q = rng.standard_normal(5).astype(dtype)
    f = o.Quadratic(Q, q)
    Lf = sv.max()
    for _ in range(100):
        x = rng.standard_normal(5).astype(dtype)
        Lest = o.lower_bound_smoothness_constant(f, x)
        assert Lest.dtype == dtype
        assert Lest <= Lf * (1 + 8 * np.finfo(dtype).eps)
    x = rng.standard_normal(5).astype(dtype)
    Lest = o.lower_bound_smoothness_constant(f, x)
    gamma_init = dtype(10) / Lest
    gamma = gamma_init
    g = o.Zero()
    for _ in range(100):
        x = rng.standard_normal(5).astype(dtype)
        # allocating variant fb_tools.jl:65-98 with A = I
        f_x, grad = f.value_and_gradient(x)
        y = x - gamma * grad
        z, g_z = g.prox(y, gamma)
        new_gamma, *_ = o.backtrack_stepsize(gamma, f, g, x, f_x, grad, y, z, g_z, x - z, alpha=0.5)
        assert new_gamma <= gamma
        gamma = new_gamma
    assert gamma < gamma_init


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fb_iterate_sequence_pin(dtype):
    """test/problems/test_equivalence.jl:51-84 pins FB's z-sequence (gamma = 0.95/||A||^2) against
    PANOC(no acceleration, 1 backtrack).  PANOC is not on the hot path; what the identity says for
    FB is that z_{k+1} = prox_{gamma g}(z_k - gamma grad f(z_k)) -- checked here directly."""
    A, b, lam, Lf = lasso_small(dtype)
    R = np.dtype(dtype).type
    gamma = R(0.95) / Lf
    it = o.ForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=np.zeros(5, dtype), gamma=gamma)
    z_prev = None
    for k, s in enumerate(itertools.islice(it, 10)):
        if z_prev is not None:
            _, grad = o.LeastSquares(A, b).value_and_gradient(z_prev)
            z_exp, _ = o.NormL1(lam).prox(z_prev - gamma * grad, gamma)
            np.testing.assert_allclose(s.z, z_exp, rtol=1e-5 if dtype == np.float32 else 1e-12, atol=1e-7)
        z_prev = s.z.copy()


def test_indbox_prox_is_clamp():
    """test/problems/test_nonconvex_qp.jl:33: z = min.(upp, max.(low, .))"""
    x = np.array([-3.0, -1.0, -0.2, 0.0, 0.7, 1.0, 5.0])
    y, v = o.IndBox(rv.NCQP_LOW, rv.NCQP_UPP).prox(x, 0.3)
    assert np.array_equal(y, np.minimum(rv.NCQP_UPP, np.maximum(rv.NCQP_LOW, x)))
    assert v == 0


@pytest.mark.parametrize("name,ffb_tol", [("lasso_tiny", 2e-4), ("lasso_small", 1e-6), ("lasso_medium", 1e-6)])
def test_shipped_benchmark_instances(name, ffb_tol):
    """benchmark/benchmarks.jl:47-61 settings (tol=1e-6, x0=0, adaptive) on the shipped data; the
    stored xstar is the pin."""
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    A, b, xstar, lam = d["A"], d["b"], d["xstar"], float(d["lam"])
    x0 = np.zeros(A.shape[1])
    z, k = o.fast_forward_backward(tol=1e-6, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam))
    assert k < 10_000
    assert np.max(np.abs(z - xstar)) <= ffb_tol
    if name != "lasso_tiny":
        z, k = o.forward_backward(tol=1e-6, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam))
        assert k < 10_000
        assert np.max(np.abs(z - xstar)) <= 1e-6


def test_synthetic_generator_reproducible_and_shardable():
    A = o.synthetic_matrix(64, 48, seed=3)
    A2 = o.synthetic_matrix(64, 48, seed=3)
    assert np.array_equal(A, A2)
    top = o.synthetic_matrix(32, 48, seed=3, row_offset=0, m_global=64)
    bot = o.synthetic_matrix(32, 48, seed=3, row_offset=32, m_global=64)
    assert np.array_equal(np.vstack([top, bot]), A)
    big = o.synthetic_matrix(512, 512, seed=0)
    assert abs(big.std() * np.sqrt(512) - 1) < 0.01 and abs(big.mean()) < 1e-3


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_douglas_rachford_pin(dtype):
    """test/problems/test_lasso_small.jl:205-214: DouglasRachford(gamma = 10/||A||^2, tol = 1e-4) with
    f = LeastSquares (prox), g = NormL1 reaches x_star in fewer than 30 iterations."""
    A, b, lam, Lf = lasso_small(dtype)
    x0 = np.zeros(5, dtype)
    y, it = o.douglas_rachford(tol=rv.LASSO_SMALL_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam),
                               gamma=dtype(10) / Lf)
    assert y.dtype == dtype
    assert np.max(np.abs(y - rv.LASSO_SMALL_XSTAR.astype(dtype))) <= rv.LASSO_SMALL_TOL
    assert it < 30
    assert np.all(x0 == 0)


def test_separable_quadratic_prox_is_the_minimiser():
    rng = np.random.default_rng(0)
    d, q, x = np.abs(rng.standard_normal(7)), rng.standard_normal(7), rng.standard_normal(7)
    f = o.SeparableQuadratic(d, q)
    y, fy = f.prox(x, 0.7)
    obj = lambda z: f(z) + np.sum((z - x) ** 2) / (2 * 0.7)
    for _ in range(50):
        assert obj(y) <= obj(y + 1e-3 * rng.standard_normal(7)) + 1e-12
    assert fy == f(y)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
class TestPANOCPins:
    def test_lasso_small_fixed_and_adaptive(self, dtype):
        """test/problems/test_lasso_small.jl:159-181: PANOC(tol=1e-4) with f = ||. - b||^2/2, A, g = NormL1"""
        A, b, lam, Lf = lasso_small(dtype)
        x0 = np.zeros(5, dtype)
        for kw, key in ((dict(Lf=Lf), "fixed"), (dict(adaptive=True), "adaptive")):
            x, it = o.panoc(tol=rv.LASSO_SMALL_TOL, x0=x0, f=o.SquaredDistance(b), A=A, g=o.NormL1(lam), **kw)
            assert x.dtype == dtype and np.max(np.abs(x - rv.LASSO_SMALL_XSTAR.astype(dtype))) <= rv.LASSO_SMALL_TOL
            assert it < rv.PANOC_LASSO_BOUNDS[key]
            assert np.all(x0 == 0)

    def test_quadratic_shortcut_equals_general_branch(self, dtype):
        """panoc.jl:215-244: the interpolation branch for generalized-quadratic f gives the same iterates as
        recomputing f and its gradient."""
        A, b, lam, Lf = lasso_small(dtype)

        class NotQuad(o.SquaredDistance):
            is_generalized_quadratic = False

        x0 = np.zeros(5, dtype)
        it1 = o.PANOCIteration(f=o.SquaredDistance(b), A=A, g=o.NormL1(lam), x0=x0, adaptive=True)
        it2 = o.PANOCIteration(f=NotQuad(b), A=A, g=o.NormL1(lam), x0=x0, adaptive=True)
        for s1, s2 in itertools.islice(zip(it1, it2), 12):
            np.testing.assert_allclose(s1.z, s2.z, rtol=2e-4 if dtype == np.float32 else 1e-9, atol=1e-6)

    def test_sparse_logistic(self, dtype):
        """test/problems/test_sparse_logistic_small.jl:101-110 (+ FB/FFB :38-73 on the composed smooth term)"""
        A = np.asfortranarray(rv.LASSO_SMALL_A.astype(dtype))
        b = rv.LASSO_SMALL_B.astype(dtype)
        lam = dtype(rv.LOGISTIC_LAM)
        xs = rv.LOGISTIC_XSTAR.astype(dtype)
        x0 = np.zeros(5, dtype)
        x, it = o.panoc(tol=rv.LOGISTIC_TOL, adaptive=True, x0=x0, f=o.LogisticLoss(b), A=A, g=o.NormL1(lam))
        assert np.max(np.abs(x - xs)) <= 1e-4 and it < rv.LOGISTIC_BOUNDS["panoc_adaptive"]
        fA = o.Composed(o.LogisticLoss(b), A)
        x, it = o.forward_backward(tol=rv.LOGISTIC_TOL, adaptive=True, x0=x0, f=fA, g=o.NormL1(lam))
        assert np.max(np.abs(x - xs)) <= 1e-4 and it < rv.LOGISTIC_BOUNDS["fb_adaptive"]
        x, it = o.fast_forward_backward(tol=rv.LOGISTIC_TOL, adaptive=True, x0=x0, f=fA, g=o.NormL1(lam))
        assert np.max(np.abs(x - xs)) <= 1e-4 and it < rv.LOGISTIC_BOUNDS["ffb_adaptive"]

    def test_fb_equals_panoc_without_acceleration(self, dtype):
        """test/problems/test_equivalence.jl:51-84: ForwardBackwardIteration == PANOCIteration(max_backtracks = 1,
        directions = NoAcceleration()) on z for 10 iterations (gamma = 0.95 / ||A||^2)."""
        A, b, lam, Lf = lasso_small(dtype)
        gamma = dtype(0.95) / Lf
        x0 = np.zeros(5, dtype)
        fb = o.ForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=x0, gamma=gamma)
        pn = o.PANOCIteration(f=o.Composed(o.SquaredDistance(b), A), A=np.eye(5, dtype=dtype), g=o.NormL1(lam), x0=x0,
                              gamma=gamma, max_backtracks=1, directions=None)
        for s_fb, s_pn in itertools.islice(zip(fb, pn), 10):
            np.testing.assert_allclose(s_fb.z, s_pn.z, rtol=1e-4 if dtype == np.float32 else 1e-8, atol=1e-6)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
class TestZeroFPRPANOCplusPins:
    def test_lasso_small(self, dtype):
        """test/problems/test_lasso_small.jl:137-157 (ZeroFPR) and :183-203 (PANOCplus): it < 20"""
        A, b, lam, Lf = lasso_small(dtype)
        x0 = np.zeros(5, dtype)
        for solver in (o.zerofpr, o.panocplus):
            for kw in (dict(Lf=Lf), dict(adaptive=True)):
                x, it = solver(tol=rv.LASSO_SMALL_TOL, x0=x0, f=o.SquaredDistance(b), A=A, g=o.NormL1(lam), **kw)
                assert x.dtype == dtype and np.max(np.abs(x - rv.LASSO_SMALL_XSTAR.astype(dtype))) <= rv.LASSO_SMALL_TOL
                assert it < 20, (solver.__name__, kw, it)

    def test_sparse_logistic(self, dtype):
        """test/problems/test_sparse_logistic_small.jl:90-99 (ZeroFPR, it < 25) and :112-121 (PANOCplus, it < 50)"""
        A = np.asfortranarray(rv.LASSO_SMALL_A.astype(dtype))
        b = rv.LASSO_SMALL_B.astype(dtype)
        xs = rv.LOGISTIC_XSTAR.astype(dtype)
        x0 = np.zeros(5, dtype)
        x, it = o.zerofpr(tol=rv.LOGISTIC_TOL, adaptive=True, x0=x0, f=o.LogisticLoss(b), A=A, g=o.NormL1(dtype(rv.LOGISTIC_LAM)))
        assert np.max(np.abs(x - xs)) <= 1e-4 and it < 25
        x, it = o.panocplus(tol=rv.LOGISTIC_TOL, adaptive=True, x0=x0, f=o.LogisticLoss(b), A=A, g=o.NormL1(dtype(rv.LOGISTIC_LAM)))
        assert np.max(np.abs(x - xs)) <= 1e-4 and it < 50

    def test_panoc_equals_panocplus(self, dtype):
        """test/problems/test_equivalence.jl:86-114: same z for 10 iterations at gamma = 0.95 / ||A||^2"""
        A, b, lam, Lf = lasso_small(dtype)
        gamma = dtype(0.95) / Lf
        x0 = np.zeros(5, dtype)
        fA = o.Composed(o.SquaredDistance(b), A)
        eye = np.eye(5, dtype=dtype)
        p1 = o.PANOCIteration(f=fA, A=eye, g=o.NormL1(lam), x0=x0, gamma=gamma)
        p2 = o.PANOCplusIteration(f=fA, A=eye, g=o.NormL1(lam), x0=x0, gamma=gamma)
        for s1, s2 in itertools.islice(zip(p1, p2), 10):
            np.testing.assert_allclose(s1.z, s2.z, rtol=2e-3 if dtype == np.float32 else 1e-7, atol=1e-5 if dtype == np.float32 else 1e-9)


# ------------------------------------------------------------------------------------------------
# second group: SFISTA, DavisYin, LiLin, DRLS, AFBA / VuCondat / ChambollePock (oracle/proxgrad_oracle_ext.py)
# ------------------------------------------------------------------------------------------------
from oracle import proxgrad_oracle_ext as ox  # noqa: E402


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
class TestSecondGroupLassoSmall:
    """test/problems/test_lasso_small.jl:205-283"""

    def xs(self, dtype):
        return rv.LASSO_SMALL_XSTAR.astype(dtype)

    def test_sfista(self, dtype):
        A, b, lam, Lf = lasso_small(dtype)
        x0 = np.zeros(5, dtype)
        y, it = ox.sfista(tol=10 * rv.LASSO_SMALL_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=Lf)
        assert y.dtype == dtype and np.all(x0 == 0)
        assert np.max(np.abs(y - self.xs(dtype))) <= 10 * rv.LASSO_SMALL_TOL
        assert it < rv.LASSO_SMALL_BOUNDS_EXT["sfista"]

    @pytest.mark.parametrize("directions,key", [("lbfgs", "drls_lbfgs"), ("nesterov_fixed", "drls_nesterov_fixed"),
                                                ("nesterov_simple", "drls_nesterov_simple")])
    def test_drls(self, dtype, directions, key):
        A, b, lam, Lf = lasso_small(dtype)
        x0 = np.zeros(5, dtype)
        z, it = ox.drls(tol=10 * rv.LASSO_SMALL_TOL, directions=directions, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam),
                        Lf=Lf)
        assert z.dtype == dtype and np.all(x0 == 0)
        assert np.max(np.abs(z - self.xs(dtype))) <= 10 * rv.LASSO_SMALL_TOL
        assert it < rv.LASSO_SMALL_BOUNDS_EXT[key]

    def test_afba(self, dtype):
        A, b, lam, Lf = lasso_small(dtype)
        R = np.dtype(dtype).type
        x0 = np.zeros(5, dtype)
        fA, g = o.LeastSquares(A, b), o.NormL1(lam)
        (x, y), it = ox.afba(theta=1, mu=1, tol=R(1e-6), x0=x0, y0=np.zeros(5, dtype), f=fA, g=g, beta_f=Lf)
        assert x.dtype == dtype and y.dtype == dtype
        assert np.max(np.abs(x - self.xs(dtype))) <= 1e-4 and it <= rv.LASSO_SMALL_BOUNDS_EXT["afba_f_g"]
        (x, y), it = ox.afba(theta=1, mu=1, tol=R(1e-6), x0=x0, y0=np.zeros(5, dtype), f=fA, h=g, beta_f=Lf)
        assert np.max(np.abs(x - self.xs(dtype))) <= 1e-4 and it <= rv.LASSO_SMALL_BOUNDS_EXT["afba_f_h"]
        (x, y), it = ox.afba(theta=1, mu=1, tol=R(1e-6), x0=x0, y0=np.zeros(4, dtype), h=ox.SqrDistance(b), L=A, g=g)
        assert np.max(np.abs(x - self.xs(dtype))) <= 1e-4 and it <= rv.LASSO_SMALL_BOUNDS_EXT["afba_h_L_g"]
        assert np.all(x0 == 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
class TestSecondGroupStronglyConvex:
    """test/problems/test_lasso_small_strongly_convex.jl:56-65, :146-153"""

    def test_sfista(self, dtype):
        A, b, lam, x0 = rv.strongly_convex_problem(dtype)
        x0b = x0.copy()
        y, it = ox.sfista(tol=rv.SC_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=rv.SC_LF, mf=rv.SC_MF)
        assert y.dtype == dtype and np.array_equal(x0, x0b)
        assert np.linalg.norm(y - rv.SC_XSTAR.astype(dtype)) <= rv.SC_TOL
        assert it < rv.SC_BOUNDS_EXT["sfista"]

    def test_drls(self, dtype):
        A, b, lam, x0 = rv.strongly_convex_problem(dtype)
        x0b = x0.copy()
        v, it = ox.drls(tol=rv.SC_TOL, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), mf=rv.SC_MF)
        assert v.dtype == dtype and np.array_equal(x0, x0b)
        assert np.max(np.abs(v - rv.SC_XSTAR.astype(dtype))) <= rv.SC_TOL
        assert it < rv.SC_BOUNDS_EXT["drls"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dr_equals_drls_without_acceleration(dtype):
    """test/problems/test_equivalence.jl:14-49"""
    A, b, lam, Lf = lasso_small(dtype)
    R = np.dtype(dtype).type
    f, g = o.LeastSquares(A, b), o.NormL1(lam)
    x0 = np.zeros(5, dtype)
    gamma = R(10) / R(np.linalg.norm(A, 2) ** 2)
    dr = iter(o.DouglasRachfordIteration(f=f, g=g, x0=x0, gamma=gamma))
    dl = iter(ox.DRLSIteration(f=f, g=g, x0=x0, gamma=gamma, lam=1, c=-np.inf, max_backtracks=1, directions="none",
                               Lf=Lf))
    for _ in range(10):
        a, bb = next(dr), next(dl)
        assert np.allclose(a.x, bb.xbar, rtol=np.sqrt(np.finfo(dtype).eps), atol=0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
class TestElasticNet:
    """test/problems/test_elasticnet.jl"""

    def test_davis_yin(self, dtype):
        A, b, _, Lf = lasso_small(dtype)
        R = np.dtype(dtype).type
        for x0 in (np.zeros(5, dtype), np.random.default_rng(3).standard_normal(5).astype(dtype)):
            x, it = ox.davis_yin(tol=R(rv.ELASTICNET_DYS["tol"]), x0=x0.copy(), f=o.LeastSquares(A, b), g=o.NormL1(R(1)),
                                 h=ox.SqrNormL2(R(1)), Lf=Lf)
            assert x.dtype == dtype
            assert np.max(np.abs(x - rv.ELASTICNET_XSTAR.astype(dtype))) <= rv.ELASTICNET_DYS["x_tol"]
            if not x0.any():
                assert it <= rv.ELASTICNET_DYS["it"]

    @pytest.mark.parametrize("theta,mu,maxit", rv.ELASTICNET_AFBA)
    def test_afba(self, dtype, theta, mu, maxit):
        A, b, _, _ = lasso_small(dtype)
        R = np.dtype(dtype).type
        rng = np.random.default_rng(5)
        for x0, y0 in ((np.zeros(5, dtype), np.zeros(4, dtype)),
                       (rng.standard_normal(5).astype(dtype), rng.standard_normal(4).astype(dtype))):
            (x, y), it = ox.afba(theta=theta, mu=mu, tol=R(1e-6), x0=x0, y0=y0, f=ox.SqrNormL2(R(1)), g=o.NormL1(R(1)),
                                 h=ox.SqrDistance(b), L=A, beta_f=1)
            assert x.dtype == dtype and y.dtype == dtype
            assert np.max(np.abs(x - rv.ELASTICNET_XSTAR.astype(dtype))) <= 1e-4
            if not x0.any():
                assert it <= maxit


def test_lilin_nonconvex_qp_tiny():
    """test/problems/test_nonconvex_qp.jl:58-66 (Float64)"""
    Q = np.diag(rv.NCQP_Q_DIAG)
    q = rv.NCQP_Q_VEC
    gamma = 0.95 / np.max(rv.NCQP_Q_DIAG)
    x0 = np.zeros(2)
    it_obj = ox.LiLinIteration(x0=x0, f=o.Quadratic(Q, q), g=o.IndBox(-1.0, 1.0), gamma=gamma)
    x, it = ox.li_lin(tol=rv.NCQP_TOL, x0=x0, f=o.Quadratic(Q, q), g=o.IndBox(-1.0, 1.0), gamma=gamma)
    z = np.minimum(1.0, np.maximum(-1.0, x - gamma * (Q @ x + q)))
    assert np.max(np.abs(x - z)) / gamma <= rv.NCQP_TOL
    assert np.all(x0 == 0) and it_obj.monitor_branch_taken == 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["anderson", "broyden"])
def test_anderson_and_broyden_operators(dtype, kind):
    """test/accel/test_anderson.jl:6-49, test/accel/test_broyden.jl:5-48"""
    R = np.dtype(dtype).type
    H, l = rv.ACCEL_H.astype(dtype), rv.ACCEL_L.astype(dtype)
    f = lambda x: R(np.dot(x, H @ x) / R(2) + np.dot(x, l))
    x_star = -np.linalg.solve(H, l)
    f_star = f(x_star)
    x = np.zeros(5, dtype)
    acc = ox.AndersonAccelerationOperator(5, x) if kind == "anderson" else ox.BroydenOperator(x)
    g = H @ x + l
    for _ in range(rv.ACCEL_ITERS):
        d = acc * g
        x = x - d
        g_prev, g = g, H @ x + l
        acc.update(-d, g - g_prev)
    assert f(x) <= f_star + (1 + abs(f_star)) * np.sqrt(np.finfo(dtype).eps)
    acc.reset()
    assert np.array_equal(acc * x, x)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["broyden", "anderson"])
def test_drls_with_broyden_and_anderson(dtype, kind):
    """test/problems/test_lasso_small.jl:216-231, rows (Broyden(), 19) and (AndersonAcceleration(5), 12)"""
    A, b, lam, Lf = lasso_small(dtype)
    x0 = np.zeros(5, dtype)
    z, it = ox.drls(tol=10 * rv.LASSO_SMALL_TOL, directions=kind, x0=x0, f=o.LeastSquares(A, b), g=o.NormL1(lam), Lf=Lf)
    assert np.max(np.abs(z - rv.LASSO_SMALL_XSTAR.astype(dtype))) <= 10 * rv.LASSO_SMALL_TOL
    assert it < rv.LASSO_SMALL_BOUNDS_EXT["drls_" + kind]


def lp_problem(dtype):
    A = np.asfortranarray(rv.LP_A.astype(dtype))
    b = A @ rv.LP_XSTAR.astype(dtype)
    c = A.T @ rv.LP_YSTAR.astype(dtype) + rv.LP_SSTAR.astype(dtype)
    return A, b, c, 100 * np.finfo(dtype).eps


def assert_lp_solution(c, A, b, x, y, tol):
    """test_linear_programs.jl:7-22 (the returned dual iterate is the negative of the LP's dual variable)"""
    assert -min(0.0, float(x.min())) <= tol
    assert np.linalg.norm(A @ x - b) <= tol
    assert max(0.0, float((-A.T @ y - c).max())) <= tol
    assert abs(float(np.dot(c + A.T @ y, x))) <= tol


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_linear_programs(dtype):
    """test/problems/test_linear_programs.jl:105-193: AFBA, VuCondat, ChambollePock, DavisYin on the LP fixture"""
    A, b, c, tol = lp_problem(dtype)
    n, m = 10, 8
    kw = dict(tol=tol, maxit=rv.LP_MAXIT, x0=np.zeros(n, dtype))
    for solver in (ox.afba, ox.vu_condat):
        (x, y), it = solver(y0=np.zeros(m, dtype), f=ox.Linear(c), g=ox.IndNonnegative(), h=ox.IndPoint(b), L=A, beta_f=0, **kw)
        assert x.dtype == dtype and y.dtype == dtype and it <= rv.LP_MAXIT
        assert_lp_solution(c, A, b, x, y, 1000 * tol)
    h = ox.SlicedSeparableSum((ox.IndPoint(b), ox.IndNonnegative()), ((0, m), (m, m + n)))
    (x, y), it = ox.chambolle_pock(y0=np.zeros(m + n, dtype), g=ox.Linear(c), h=h, L=np.vstack([A, np.eye(n, dtype=dtype)]), **kw)
    assert it <= rv.LP_MAXIT
    assert_lp_solution(c, A, b, x, y[:m], 1000 * tol)
    xf, it = ox.davis_yin(gamma=dtype(1), f=ox.Linear(c), g=ox.IndNonnegative(), h=ox.IndAffine(A, b), **kw)
    assert xf.dtype == dtype and it <= rv.LP_MAXIT
    assert np.linalg.norm(xf - rv.LP_XSTAR.astype(dtype)) <= 100 * tol


def test_cpu_twin_matches_the_numpy_oracle():
    """oracle/csrc/cpu_twin.c -- the C / OpenMP restatement bench.py times as its CPU leg -- against the numpy oracle
    (fast_forward_backward.jl:73-145 restated, itself pinned to the reference's known answers above): same iterates after
    25 fixed-step iterations up to summation order, for every thread count (the partial sums of A x are combined in thread
    order)."""
    from oracle import cpu_twin

    m, n = 300, 900
    A, b, _ = o.synthetic_lasso(m, n, seed=2, dtype=np.float32)
    A = np.asfortranarray(A)
    lam = np.float32(0.1) * np.float32(np.max(np.abs(A.T @ b)))
    Lf = np.float32(1.05) * np.float32(np.linalg.norm(A.astype(np.float64), 2) ** 2)
    it = iter(o.FastForwardBackwardIteration(f=o.LeastSquares(A, b), g=o.NormL1(lam), x0=np.zeros(n, np.float32), Lf=Lf))
    for _ in range(26):
        s = next(it)
    for threads in (1, 3, None):
        z, fx, sec, thr = cpu_twin.ffb(A, b, lam, Lf, 25, threads=threads)
        assert np.max(np.abs(z - s.z)) <= 1e-5 * max(1.0, float(np.max(np.abs(s.z))))
        assert abs(fx - float(s.f_x)) <= 1e-5 * abs(float(s.f_x)) and sec > 0 and thr >= 1
        assert np.array_equal(z != 0, s.z != 0) or np.count_nonzero((z != 0) != (s.z != 0)) <= 2
    # the host read-ceiling pass of the CPU leg: every element is read (the sum says so), the rate is a positive number
    import ctypes as C

    big = np.arange(200_003, dtype=np.float32) % 7
    sec = C.c_double()
    for fn in (cpu_twin.load().cpu_twin_read_pass, cpu_twin.load().cpu_twin_read_pass_seq):
        total = fn(big.ctypes.data, big.size, 3, C.byref(sec))
        assert total == 3 * float(big.astype(np.float64).sum()) and sec.value > 0
    assert cpu_twin.read_gbps(big, threads=2, min_seconds=0.01) > 0
