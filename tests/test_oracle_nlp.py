"""the full-NLP formulation of python/mpc.py:116-216 (``ihm2_amd/mpc.py::nlp_ocp``).  CPU side: the oracle's
SQP run to convergence on that programme against SciPy's SLSQP on the SAME programme written out independently (decision vector, cost
without reference, boxes, shooting equalities through the oracle's integrator, rate rows) -- pins the problem statement, not the solver."""
import numpy as np
import pytest

from conftest import sample_x0

NF, DT = 3, 0.05


@pytest.fixture(scope="module")
def problem(track):
    from ihm2_amd import mpc, ocp as O
    from oracle import oracle as orc

    Q = np.diag([0.0, 4.0, 2.0, 0.3, 0.1, 0.1, 1e-5, 1.0]); Q[1, 2] = Q[2, 1] = 0.5
    R = np.diag([1e-5, 2.0]); Qf = 3.0 * Q
    mb = mpc.ModelBounds(v_x_min=1.0, delta_max=0.3, delta_dot_max=2.0)
    ocp = mpc.nlp_ocp(O.fkin6_model, NF, mb, DT, 2 * track.s_ref.size, Q, R, Qf, max_iter=60, tol=1e-7)
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, 3, seed=11)[1]
    x0[1], x0[2], x0[3] = 0.8, 0.12, 9.0
    return dict(ocp=ocp, P=P, Q=Q, R=R, Qf=Qf, mb=mb, x0=x0, orc=orc)


def _sqp(problem):
    P, x0 = problem["P"], problem["x0"]
    x = np.tile(x0, (1, NF + 1, 1)).copy(); u = np.tile(x0[6:8], (1, NF, 1)).copy()
    out = P.sqp_solve(x, u, x0[None], np.zeros((1, NF, 12)), np.zeros((1, 8)), max_iter=60, tol=1e-7)
    return x[0], u[0], out


def test_converged_sqp_satisfies_the_programme(problem):
    x, u, out = _sqp(problem)
    assert out["status"][0] == 0 and np.all(out["res"][0] <= 1e-7)
    mb, P = problem["mb"], problem["P"]
    assert np.all(np.abs(x[:, 1]) <= mb.n_max + 1e-9) and np.all(x[:, 3] >= mb.v_x_min - 1e-9) and np.all(np.abs(x[:, 7]) <= mb.delta_max + 1e-9)
    assert np.all(np.abs(u[:, 1] - x[:-1, 7]) <= 0.02 * mb.delta_dot_max + 1e-9)          # rate rows (t_delta = 0.02)
    xn = P.sim_step(x[:-1], u, 0, 25)
    assert np.max(np.abs(xn - x[1:])) < 1e-8                                              # shooting equalities


def test_converged_sqp_is_the_slsqp_solution_of_the_written_out_programme(problem):
    from scipy.optimize import minimize

    x, u, _ = _sqp(problem)
    P, Q, R, Qf, mb, x0 = (problem[k] for k in ("P", "Q", "R", "Qf", "mb", "x0"))
    ns = 8 * (NF + 1)

    def split(w):
        return w[:ns].reshape(NF + 1, 8), w[ns:].reshape(NF, 2)

    def cost(w):
        X, U = split(w)
        return float(np.einsum("ki,ij,kj->", X[:-1], Q, X[:-1]) + np.einsum("ki,ij,kj->", U, R, U) + X[-1] @ Qf @ X[-1])

    def shooting(w):
        X, U = split(w)
        return np.concatenate(((P.sim_step(X[:-1].copy(), U.copy(), 0, 25) - X[1:]).reshape(-1), X[0] - x0))

    def rates(w):                                     # ulin - |x_act - u| >= 0
        X, U = split(w)
        d = X[:-1, 6:8] - U
        lim = np.array([1e-3 * mb.T_dot_max, 0.02 * mb.delta_dot_max])
        return np.concatenate(((lim - d).reshape(-1), (lim + d).reshape(-1)))

    lo = np.full((NF + 1, 8), -np.inf); hi = np.full((NF + 1, 8), np.inf)
    lo[:, [1, 3, 6, 7]] = [-mb.n_max, mb.v_x_min, -mb.T_max, -mb.delta_max]
    hi[:, [1, 3, 6, 7]] = [mb.n_max, mb.v_x_max, mb.T_max, mb.delta_max]
    bounds = list(zip(lo.reshape(-1), hi.reshape(-1))) + [(-mb.T_max, mb.T_max), (-mb.delta_max, mb.delta_max)] * NF
    w0 = np.concatenate((x.reshape(-1), u.reshape(-1)))
    # start SLSQP away from the SQP point: it has to come back to it
    rng = np.random.default_rng(5)
    w_start = w0 + 1e-2 * rng.standard_normal(w0.size) * np.maximum(1.0, np.abs(w0))
    r = minimize(cost, w_start, method="SLSQP", bounds=bounds, constraints=[{"type": "eq", "fun": shooting}, {"type": "ineq", "fun": rates}],
                 options=dict(maxiter=500, ftol=1e-14))
    assert r.success, r.message
    assert abs(r.fun - cost(w0)) <= 1e-6 * max(1.0, abs(r.fun))                    # same optimal value ...
    Xs, Us = split(r.x)
    # ... same point: 2e-4 relative on every component the cost determines; the torque state and input carry the weight 1e-5 (Q[6,6], R[0,0]:
    # the objective is flat along them, 1e-3 relative is what two converged solvers agree to -- the optimal VALUE above pins them)
    ex = np.abs(Xs - x) / np.maximum(1.0, np.abs(x)); eu = np.abs(Us - u) / np.maximum(1.0, np.abs(u))
    stiff = [0, 1, 2, 3, 4, 5, 7]
    assert ex[:, stiff].max() < 2e-4 and eu[:, 1].max() < 2e-4 and ex[:, 6].max() < 1e-3 and eu[:, 0].max() < 1e-3
