"""Known-answer DATA transcribed from the reference's own test-suite (inputs and expected
outputs only -- no reference code).  Each block cites where the numbers come from
(paths relative to /root/reference)."""
import numpy as np

# ---- test/problems/test_lasso_small.jl:17-23, 42, 44 (also test_equivalence.jl:52-58) ----
LASSO_SMALL_A = np.array(
    [
        [1.0, -2.0, 3.0, -4.0, 5.0],
        [2.0, -1.0, 0.0, -1.0, 3.0],
        [-1.0, 0.0, 4.0, -3.0, 2.0],
        [-1.0, -1.0, -1.0, 1.0, 3.0],
    ]
)
LASSO_SMALL_B = np.array([1.0, 2.0, 3.0, 4.0])
LASSO_SMALL_XSTAR = np.array([-3.877278911564627e-01, 0, 0, 2.174149659863943e-02, 6.168435374149660e-01])
LASSO_SMALL_TOL = 1e-4
# lam = 0.1 * norm(A' * b, Inf)   (test_lasso_small.jl:29)    Lf = opnorm(A)^2 (:40)
# iteration-count upper bounds asserted by the reference (test_lasso_small.jl)
LASSO_SMALL_BOUNDS = {
    "fb_fixed": 150,  # :53
    "fb_adaptive": 300,  # :64
    "fb_adaptive_regret": 150,  # :79   increase_gamma = 1.01
    "ffb_fixed": 100,  # :90
    "ffb_adaptive": 200,  # :101
    "ffb_adaptive_regret": 100,  # :116
    "ffb_fixed_custom_seq": 100,  # :133  FixedNesterovSequence
}

# ---- test/problems/test_lasso_small_strongly_convex.jl:10-54 ----
SC_DIM = 5
SC_MF = 1.0
SC_LF = 10.0
SC_XSTAR = np.array(
    [0.8466800540711814, 0.17674262101590932, -0.4987234606672925, 0.5531315167924573, -0.14739365562631113]
)
SC_W = np.array([0.15823052457732423, 0.6874613398393697, 0.9357764685973888, 0.05863707298785681, 0.49087050154723844])
SC_B = np.array(
    [
        [0.6997086717991196, 0.37124544422925876, 0.31840520080247225, 0.20097960566711592, 0.038329117953706526],
        [0.1134636504826555, 0.8273912343075426, 0.8997522727456534, 0.9821118072706589, 0.9100659142463259],
        [0.9701886480567284, 0.42825250593295605, 0.6952640061565183, 0.9699979632534245, 0.6106722979088736],
        [0.4442755181780246, 0.4641748710746476, 0.9716060376558348, 0.5951146731055232, 0.5699044913634803],
        [0.6681510415197733, 0.35423403325449887, 0.28461925562068024, 0.15941152427241456, 0.6499046326711716],
    ]
)
SC_TOL = 1e-4
SC_BOUNDS = {
    "fb_fixed": 110,  # :70
    "fb_adaptive": 300,  # :79
    "fb_adaptive_regret": 80,  # :92
    "ffb_fixed_mf": 35,  # :101
    "ffb_adaptive": 100,  # :110
    "ffb_adaptive_regret": 100,  # :123
    "ffb_constant_seq": 35,  # :142
}


def strongly_convex_problem(dtype):
    """Construction of test_lasso_small_strongly_convex.jl:22-44 (data recipe):
    lam = (mf + Lf)/2; D = Diagonal(sqrt(mf) .+ (sqrt(Lf) - sqrt(mf)) * w); D[1] = sqrt(mf);
    D[end] = sqrt(Lf); Q = qr(B).Q; A = Q D Q'; b = A x_star + lam inv(A') sign.(x_star);
    x0 = A \\ b."""
    T = np.dtype(dtype).type
    mf, Lf = T(SC_MF), T(SC_LF)
    lam = (mf + Lf) / T(2)
    d = (np.sqrt(mf) + (np.sqrt(Lf) - np.sqrt(mf)) * SC_W.astype(dtype)).astype(dtype)
    d[0] = np.sqrt(mf)
    d[-1] = np.sqrt(Lf)
    Q, _ = np.linalg.qr(SC_B.astype(dtype))
    A = (Q * d[None, :]) @ Q.T
    xs = SC_XSTAR.astype(dtype)
    b = A @ xs + lam * (np.linalg.inv(A.T) @ np.sign(xs))
    x0 = np.linalg.solve(A, b)
    return np.asfortranarray(A.astype(dtype)), b.astype(dtype), lam, x0.astype(dtype)


# ---- test/accel/test_lbfgs.jl:6-101 ----
LBFGS_Q = np.array(
    [
        [32.0, 13.1, -4.9, -3.0, 6.0, 2.2, 2.6, 3.4, -1.9, -7.5],
        [13.1, 18.3, -5.3, -9.5, 3.0, 2.1, 3.9, 3.0, -3.6, -4.4],
        [-4.9, -5.3, 7.7, 2.1, -0.4, -3.4, -0.8, -3.0, 5.3, 5.5],
        [-3.0, -9.5, 2.1, 20.1, 1.1, 0.8, -12.4, -2.5, 5.5, 2.1],
        [6.0, 3.0, -0.4, 1.1, 3.8, 0.6, 0.5, 0.9, -0.4, -2.0],
        [2.2, 2.1, -3.4, 0.8, 0.6, 7.8, 2.9, -1.3, -4.3, -5.1],
        [2.6, 3.9, -0.8, -12.4, 0.5, 2.9, 14.5, 1.7, -4.9, 1.2],
        [3.4, 3.0, -3.0, -2.5, 0.9, -1.3, 1.7, 6.6, -0.8, 2.7],
        [-1.9, -3.6, 5.3, 5.5, -0.4, -4.3, -4.9, -0.8, 7.9, 5.7],
        [-7.5, -4.4, 5.5, 2.1, -2.0, -5.1, 1.2, 2.7, 5.7, 16.1],
    ]
)
LBFGS_q = np.array([2.9, 0.8, 1.3, -1.1, -0.5, -0.3, 1.0, -0.3, 0.7, -2.1])
LBFGS_XS = np.array(
    [
        [1.0, 0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07, 0.08, 0.09],
        [0.09, 1.0, 0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07, 0.08],
        [0.08, 0.09, 1.0, 0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07],
        [0.07, 0.08, 0.09, 1.0, 0.01, 0.02, 0.03, 0.04, 0.05, 0.06],
        [0.06, 0.07, 0.08, 0.09, 1.0, 0.01, 0.02, 0.03, 0.04, 0.05],
    ]
)
LBFGS_MEM = 3  # test_lbfgs.jl:104
LBFGS_DIRS_REF = np.array(
    [
        [
            -3.476000000000000e01, -1.367700000000000e01, 2.961000000000000e00, 3.756000000000000e00,
            -5.618000000000001e00, -1.571000000000000e00, -4.121000000000000e00, -3.709000000000000e00,
            4.010000000000000e-01, 7.639999999999999e00,
        ],
        [
            -6.861170733797231e-01, -1.661270665201917e00, 2.217225828759783e-01, 5.615134140894827e-01,
            -1.922426760799171e-01, -8.961101045874649e-02, -3.044802963260585e-01, -1.996235459345302e-01,
            1.267604425710271e-01, 3.360845247013288e-01,
        ],
        [
            -1.621334774299757e-01, 2.870743130038511e-01, -5.485761164147891e-01, 9.992734938824949e-02,
            -1.332550298134261e-02, 5.326252573648003e-02, -6.299408068289100e-02, 1.525398352758626e-02,
            -7.776943954825602e-02, -2.335884953507600e-02,
        ],
        [
            -2.008976150849174e-01, 2.237224648542354e-01, 4.811889625788801e-02, -6.855884193567087e-01,
            -2.729265954345345e-02, 3.651730112313705e-02, 6.325330777317102e-02, 2.871281112230844e-02,
            -1.285590864125103e-01, -3.204963735369062e-03,
        ],
        [
            -2.317011191832649e-01, 2.980080835636926e-02, -1.267017945785352e-01, 4.328230970765587e-02,
            -2.437461022925742e-01, 1.349716200511426e-02, -7.155992987801297e-04, -3.513449694839536e-03,
            -5.603489763638488e-02, 5.612114259243499e-02,
        ],
    ]
)

# ---- test/accel/test_nesterov.jl:15-22 (5x5 quadratic for the Beck-Teboulle bound) ----
NESTEROV_H = np.array(
    [
        [0.63287, 0.330934, -0.156908, -0.294776, 0.10761],
        [0.330934, 0.673201, 0.0459778, 0.231011, -0.235265],
        [-0.156908, 0.0459778, 0.635812, -0.232261, -0.388775],
        [-0.294776, 0.231011, -0.232261, 0.726854, -0.0691783],
        [0.10761, -0.235265, -0.388775, -0.0691783, 0.336262],
    ]
)
NESTEROV_l = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
NESTEROV_FIXED_GAMMA = 1.7  # test_nesterov.jl:65

# ---- test/problems/test_nonconvex_qp.jl:10-24 ----
NCQP_Q_DIAG = np.array([-0.5, 1.0])
NCQP_q = np.array([0.3, 0.5])
NCQP_LOW, NCQP_UPP = -1.0, 1.0

# ---- test/problems/test_sparse_logistic_small.jl:8-36 (A, b as LASSO_SMALL_*; lam = 0.1; TOL = 1e-6) ----
LOGISTIC_XSTAR = np.array([0, 0, 2.114635341704963e-01, 0, 2.845881348733116e00])
LOGISTIC_LAM = 0.1
LOGISTIC_TOL = 1e-6
LOGISTIC_BOUNDS = {"fb_adaptive": 1100, "fb_adaptive_regret": 500, "ffb_adaptive": 500, "ffb_adaptive_regret": 200,
                   "panoc_adaptive": 50}  # :45,60,71,86,108
PANOC_LASSO_BOUNDS = {"fixed": 20, "adaptive": 20}  # test_lasso_small.jl:167,179


# ---- second group of pins: SFISTA / DRLS / AFBA on test_lasso_small.jl, DavisYin / AFBA on test_elasticnet.jl ----
LASSO_SMALL_BOUNDS_EXT = {
    "dr": 30,  # test_lasso_small.jl:212  DouglasRachford(gamma = 10 / opnorm(A)^2)
    "drls_lbfgs": 17,  # :217  DRLS(tol = 10 TOL, directions = LBFGS(5)), x within 10 TOL
    "drls_nesterov_fixed": 36,  # :220
    "drls_nesterov_simple": 36,  # :221
    "afba_f_g": 80,  # :247   AFBA(theta = 1, mu = 1, tol = 1e-6)(f = fA, g = g, beta_f = opnorm(A)^2), x within 1e-4
    "afba_f_h": 100,  # :261  same with h = g instead of g
    "afba_h_L_g": 150,  # :270  (h = f_prox, L = A, g = g), y0 in R^m
    "sfista": 100,  # :281  SFISTA(tol = 10 TOL), x within 10 TOL
}
SC_BOUNDS_EXT = {"sfista": 40, "drls": 14}  # test_lasso_small_strongly_convex.jl:61, :151

# test_elasticnet.jl:8-29: same A, b as the small LASSO; reg = NormL1(1) + SqrNormL2(1); loss = ||. - b||^2 / 2
ELASTICNET_XSTAR = np.array([-0.6004983388704322, 0.0, 0.0, 0.195182724252491, 0.764119601328903])
ELASTICNET_DYS = {"tol": 1e-6, "x_tol": 1e-3, "it": 140}  # :37-41 (<=)
ELASTICNET_AFBA = [(2, 0, 130), (1, 1, 2000), (0, 1, 320), (0, 0, 194), (1, 0, 130)]  # :56 (theta, mu, maxit); x within 1e-4

# test_nonconvex_qp.jl:9-31 (tiny): Q = Diagonal(-0.5, 1), q = (0.3, 0.5), box [-1, 1], gamma = 0.95 / max(diag Q)
NCQP_Q_VEC = np.array([0.3, 0.5])
NCQP_TOL = 1e-4

# ---- test/accel/test_anderson.jl:7-16 and test_broyden.jl:6-15 (same data): quadratic f(x) = <x, Hx>/2 + <x, l> ----
ACCEL_H = np.array(
    [
        [0.63287, 0.330934, -0.156908, -0.294776, 0.10761],
        [0.330934, 0.673201, 0.0459778, 0.231011, -0.235265],
        [-0.156908, 0.0459778, 0.635812, -0.232261, -0.388775],
        [-0.294776, 0.231011, -0.232261, 0.726854, -0.0691783],
        [0.10761, -0.235265, -0.388775, -0.0691783, 0.336262],
    ]
)
ACCEL_L = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
ACCEL_ITERS = 10  # :33  after 10 quasi-Newton steps f(x) <= f_star + (1 + |f_star|) sqrt(eps(R))
LASSO_SMALL_BOUNDS_EXT.update({"drls_broyden": 19, "drls_anderson": 12})  # test_lasso_small.jl:218-219
