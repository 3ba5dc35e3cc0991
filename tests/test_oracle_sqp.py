"""Oracle, globalised SQP (python/main.py:230-237: "SQP", max_iter 2, "MERIT_BACKTRACKING"): self-certifying checks of the
line search -- acados is absent, so these pin the algorithm's own invariants (PARITY UNPINNED, see oracle/ihm2_oracle.h)."""
import numpy as np
from conftest import make_ocp, sample_x0

from oracle import oracle as orc
from test_oracle_rti import _stanley_guess

N = 40


def _problem(track, B, seed=5, **opts):
    ocp = make_ocp(qp_tol=1e-8, qp_solver_iter_max=60, **opts)
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, B, seed=seed)
    x0[:, 5] = x0[:, 3] * np.interp(x0[:, 0], track.s_ref, track.kappa_ref)
    x, u = _stanley_guess(P, track, x0)
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(N)[None] / N
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
    return P, x0, x, u, yref, yref_e


def test_fixed_step_sqp_is_repeated_rti(track):
    P, x0, x, u, yref, yref_e = _problem(track, 4)
    xa, ua = x.copy(), u.copy()
    out = P.sqp_solve(xa, ua, x0, yref, yref_e, max_iter=3, globalization="FIXED_STEP", tol=0.0)
    pi = lam = None
    for _ in range(3):
        r = P.rti_step(x, u, x0, yref, yref_e, pi=pi, lam=lam)
        pi, lam = r["pi"], r["lam"]
    np.testing.assert_array_equal(xa, x); np.testing.assert_array_equal(ua, u)
    np.testing.assert_array_equal(out["pi"], pi); np.testing.assert_array_equal(out["lam"], lam)
    assert np.all(out["status"] == 2) and np.all(out["sqp_iter"] == 3) and np.all(out["alpha"] == 1.0)   # ACADOS_MAXITER


def test_a_converged_iterate_is_left_alone(track):
    P, x0, x, u, yref, yref_e = _problem(track, 6)
    x0[:, 1] *= 0.4; x0[:, 2] *= 0.3; x0[:, 3] = 14 + 0.25 * x0[:, 3]; x0[:, 6] = 150.0
    x0[:, 5] = x0[:, 3] * np.interp(x0[:, 0], track.s_ref, track.kappa_ref)
    x, u = _stanley_guess(P, track, x0)
    out = P.sqp_solve(x, u, x0, yref, yref_e, max_iter=25, tol=[1e-3, 1e-8, 1e-8, 1e-3])
    conv = out["status"] == 0
    assert conv.sum() >= 3 and np.all(out["sqp_iter"][conv] < 25)
    xs, us = x.copy(), u.copy()
    again = P.sqp_solve(x, u, x0, yref, yref_e, pi=out["pi"], lam=out["lam"], sl=out["sl"], max_iter=5, tol=[1e-3, 1e-8, 1e-8, 1e-3])
    assert np.all(again["status"][conv] == 0) and np.all(again["sqp_iter"][conv] == 0)
    np.testing.assert_array_equal(x[conv], xs[conv]); np.testing.assert_array_equal(u[conv], us[conv])


def test_backtracking_picks_steps_from_the_ladder_and_never_raises_the_merit(track):
    B = 24
    P, x0, x, u, yref, yref_e = _problem(track, B, seed=11)
    u[:, :, 1] += 0.3 * np.sin(np.arange(N))[None]          # a poor steering guess: full steps overshoot
    ladder = np.concatenate([0.7 ** np.arange(9), [0.05]])
    pi = lam = sl = None
    n_short = 0
    for it in range(6):
        xp, up = x.copy(), u.copy()
        out = P.sqp_solve(x, u, x0, yref, yref_e, pi=pi, lam=lam, sl=sl, max_iter=1)
        pi, lam, sl = out["pi"], out["lam"], out["sl"]
        assert np.all(np.isin(out["status"], (0, 2)))
        moved = out["sqp_iter"] == 1
        assert np.all(np.min(np.abs(out["alpha"][moved, None] - ladder[None]), axis=1) < 1e-12)
        n_short += int((out["alpha"][moved] < 1.0).sum())
        # the accepted point is the convex combination of the start and the full step of a plain RTI iteration
        xf, uf = xp.copy(), up.copy()
        P.rti_step(xf, uf, x0, yref, yref_e, pi=None if it == 0 else pi_prev, lam=None if it == 0 else lam_prev)
        a = out["alpha"][:, None, None]
        np.testing.assert_allclose(x[moved], (xp + a * (xf - xp))[moved], rtol=0, atol=1e-12)
        np.testing.assert_allclose(u[moved], (up + a * (uf - up))[moved], rtol=0, atol=1e-9)
        pi_prev, lam_prev = pi.copy(), lam.copy()
    assert n_short >= 3                                       # the test is not vacuous
    assert np.all(np.isfinite(x)) and np.all(np.isfinite(u))
    # Armijo variant: never a longer step than plain decrease allows on the same problem
    P2, x0, x, u, yref, yref_e = _problem(track, B, seed=11)
    u[:, :, 1] += 0.3 * np.sin(np.arange(N))[None]
    xa, ua = x.copy(), u.copy()
    plain = P2.sqp_solve(x, u, x0, yref, yref_e, max_iter=1)
    armijo = P2.sqp_solve(xa, ua, x0, yref, yref_e, max_iter=1, use_sufficient_descent=True)
    assert np.all(armijo["alpha"] <= plain["alpha"] + 1e-15)


def test_line_search_helps_from_a_poor_guess(track):
    """Full steps from a poor guess leave more dynamics defect after a few iterations than damped ones, in the median."""
    B = 32
    P, x0, x, u, yref, yref_e = _problem(track, B, seed=3)
    u[:, :, 1] += 0.3 * np.sin(np.arange(N))[None]
    xa, ua = x.copy(), u.copy()
    full = P.sqp_solve(x, u, x0, yref, yref_e, max_iter=4, globalization="FIXED_STEP", tol=0.0)
    damped = P.sqp_solve(xa, ua, x0, yref, yref_e, max_iter=4, tol=0.0)
    assert np.all(np.isin(full["status"], (2, 4))) and np.all(np.isin(damped["status"], (2, 4)))
    assert (damped["status"] == 2).sum() >= (full["status"] == 2).sum()
