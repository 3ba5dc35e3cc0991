"""Oracle SQP-RTI: repeated iterations on a fixed problem converge to a KKT point of the NLP
(self-certifying, SURVEY.md 8c check 6), controller shift semantics, failure isolation."""
import numpy as np
from conftest import make_ocp, sample_x0

from oracle import oracle as orc

N = 40


def _stanley_guess(P, track, x0, M=25):
    return orc.stanley_guess(P, track.s_ref, track.kappa_ref, x0, N, M)


def test_sqp_iterations_reach_kkt_points(track):
    """Repeated full-step RTI iterations on a frozen problem: every instance that settles does so at
    a KKT point of the NLP (stationarity at rounding level relative to |grad| ~ 1e4, defects and
    bound violation below the QP tolerance).  Undamped Gauss-Newton SQP on this non-convex problem may also enter
    a 2-cycle (rate-limited bang-bang steering); such instances are not counted, they only must
    stay finite with status 0."""
    ocp = make_ocp(qp_tol=1e-8, qp_solver_iter_max=60)
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    B = 6
    x0 = sample_x0(track, B, seed=5)
    x0[:, 1] *= 0.4; x0[:, 2] *= 0.3; x0[:, 3] = 14 + 0.25 * x0[:, 3]; x0[:, 6] = 150.0
    x0[:, 5] = x0[:, 3] * np.interp(x0[:, 0], track.s_ref, track.kappa_ref)
    x, u = _stanley_guess(P, track, x0)
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(N)[None] / N
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
    pi = lam = None
    hist = []
    for it in range(12):
        out = P.rti_step(x, u, x0, yref, yref_e, pi=pi, lam=lam)
        pi, lam = out["pi"], out["lam"]
        assert np.all(out["status"] == 0)
        hist.append(out["res"].copy())
    hist = np.array(hist)            # (it, B, 4)
    final = hist[-1]
    gscale = 1e4
    settled = final[:, 1] < 1e-10
    assert settled.sum() >= 3, final
    assert np.all(final[settled, 0] < 1e-8 * gscale), final[:, 0]     # stationarity
    assert np.all(final[settled, 2] < 1e-8), final[:, 2]              # bound violation <= qp_tol (slack residual of the IPM)
    assert np.all(final[settled, 3] < 1e-7 * gscale), final[:, 3]     # complementarity ~ qp_tol * |g|
    assert np.all(hist[6, settled, 1] < 1e-6 * hist[1, settled, 1])    # fast local contraction
    assert np.all(np.abs(x[:, 0] - x0) < 1e-9) and np.all(np.isfinite(x)) and np.all(np.isfinite(u))


def test_prepare_step_matches_controller_semantics():
    """python/main.py:297-322."""
    rng = np.random.default_rng(0)
    B, n = 3, 7
    x = rng.normal(size=(B, n + 1, 8)); u = rng.normal(size=(B, n, 2)); x0 = rng.normal(size=(B, 8))
    xp, up = x.copy(), u.copy()
    yref, yref_e = orc.prepare_step(n, x0, 40.0, x, u)
    for j in range(n - 1):
        np.testing.assert_array_equal(x[:, j], xp[:, j + 1])
        np.testing.assert_array_equal(u[:, j], up[:, j + 1])
    np.testing.assert_array_equal(x[:, n - 1], xp[:, n])
    np.testing.assert_array_equal(x[:, n], xp[:, n])
    assert np.all(u[:, n - 1] == 0)
    for j in range(n):
        np.testing.assert_allclose(yref[:, j, 0], x0[:, 0] + 40.0 * j / n, rtol=1e-15)
        assert np.all(yref[:, j, 1:] == 0)
    np.testing.assert_allclose(yref_e[:, 0], x0[:, 0] + 40.0, rtol=1e-15)
    assert np.all(yref_e[:, 1:] == 0)


def test_failed_instance_does_not_poison_batch(track):
    ocp = make_ocp()
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, 4, seed=9)
    x, u = _stanley_guess(P, track, x0)
    x0_bad = x0.copy()
    x0_bad[1, 1] = 30.0        # 30 m off the centre line: |n| <= 2 cannot be met at stage 1
    yref = np.zeros((4, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(N)[None] / N
    yref_e = np.zeros((4, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
    xg, ug = x.copy(), u.copy()
    out_bad = P.rti_step(xg, ug, x0_bad, yref, yref_e)
    xr, ur = x.copy(), u.copy()
    out_ref = P.rti_step(xr, ur, x0, yref, yref_e)
    assert out_bad["status"][1] == 4
    np.testing.assert_array_equal(xg[1], x[1])                 # failed instance keeps its iterate
    for b in (0, 2, 3):
        assert out_bad["status"][b] == 0
        np.testing.assert_array_equal(xg[b], xr[b])            # batch-mates are untouched
    assert np.all(np.isfinite(xg)) and np.all(np.isfinite(ug))
