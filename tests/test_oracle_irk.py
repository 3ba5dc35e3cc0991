"""The oracle's IRK integrators (acados IRK: 4-stage Gauss-Legendre for the OCP, python/main.py:234-236; Radau IIA for the plants,
python/main.py:395-400): against scipy's Radau at rtol 1e-12, orders of convergence, the stability function on the stiff actuator
lag, sensitivities against central differences and against RK4 x 400."""
import numpy as np
import pytest
from conftest import random_state

from oracle import oracle as orc


@pytest.fixture(scope="module")
def tab(track):
    return track.s_ref, track.kappa_ref


def _ref(model, x, u, tab, dt):
    from scipy.integrate import solve_ivp

    sol = solve_ivp(lambda t, y: orc.f(model, y, u, *tab), [0, dt], x, method="Radau", rtol=1e-12, atol=1e-14)
    return sol.y[:, -1]


@pytest.mark.parametrize("model", [orc.MODEL_FKIN6, orc.MODEL_FDYN6U])
@pytest.mark.parametrize("smooth", [True, False])
def test_irk_matches_scipy_radau(tab, model, smooth):
    """Tolerance 1e-10 relative on a curvature table without interior knots (kappa linear in s: the flow is smooth); on the real
    table kappa(s) is piecewise linear -- every step that crosses a knot falls back to low order, and two fine runs (and scipy's
    adaptive Radau) agree to ~2e-10 only, so the bar there is 2e-9."""
    if smooth:
        tab = (np.array([-400.0, 800.0]), np.array([-0.02, 0.06]))
    tol = 1e-10 if smooth else 2e-9
    rng = np.random.default_rng(5)
    for _ in range(4):
        x, u = random_state(rng)
        x[3] = rng.uniform(4, 15)
        ref = _ref(model, x, u, tab, 0.05)
        # the plants' integrator (Radau IIA x 100 steps) and a fine Gauss-Legendre run
        xs = [orc.rk4(model, x, u, *tab, 0.05, 100, integrator=integ) for integ in (orc.INTEG_IRK_RADAU4, orc.INTEG_IRK_GL4)]
        assert np.max(np.abs(xs[0] - xs[1]) / (1 + np.abs(xs[1]))) < tol
        for xn in xs:
            assert np.max(np.abs(xn - ref) / (1 + np.abs(ref))) < tol


def test_irk_converges_with_the_step_count(tab):
    """One step per interval (the reference's setting) is already accurate to 1e-4 on the non-stiff states -- three Newton iterations
    from K = 0 and the kinks of kappa(s) bound it, not the order of the collocation -- and the error falls with M."""
    rng = np.random.default_rng(7)
    x, u = random_state(rng); x[3] = 9.0
    x[6], u[0] = 100.0, 100.0          # the throttle lag at rest (its z = -50 per interval is the subject of the next test)
    ref = orc.rk4(orc.MODEL_FKIN6, x, u, *tab, 0.05, 800, integrator=orc.INTEG_IRK_RADAU4)
    for integ in (orc.INTEG_IRK_GL4, orc.INTEG_IRK_RADAU4):
        e = [np.max(np.abs(orc.rk4(orc.MODEL_FKIN6, x, u, *tab, 0.05, M, integrator=integ) - ref)) for M in (1, 2, 4, 8)]
        assert e[0] < 2e-4 and e[3] < 1e-7 and e[0] > e[1] > e[2] > e[3], (integ, e)
    # against RK4 x 25 (the ERK configuration of this build): same flow, agreement at the level of the coarser of the two
    xe = orc.rk4(orc.MODEL_FKIN6, x, u, *tab, 0.05, 25)
    xi = orc.rk4(orc.MODEL_FKIN6, x, u, *tab, 0.05, 1, integrator=orc.INTEG_IRK_GL4)
    assert np.max(np.abs(xe - xi)) < 2e-4


def test_stability_functions_on_the_throttle_lag(tab):
    """T' = (u_T - T) / t_T, z = -dt / t_T = -50: one step multiplies T - u_T by R(z).
    Gauss-Legendre 4: R = Pade(4,4) -- A-stable, |R(-50)| = 0.45; Radau IIA: L-stable (R -> 0 as z -> -inf), |R(-50)| = 0.043."""
    x = np.array([10.0, 0.0, 0.0, 8.0, 0.0, 0.0, 100.0, 0.0]); u = np.array([150.0, 0.0])
    z = -50.0
    num = 1 + z / 2 + 3 * z**2 / 28 + z**3 / 84 + z**4 / 1680
    den = 1 - z / 2 + 3 * z**2 / 28 - z**3 / 84 + z**4 / 1680
    xn = orc.rk4(orc.MODEL_FKIN6, x, u, *tab, 0.05, 1, integrator=orc.INTEG_IRK_GL4)
    assert (xn[6] - 150.0) / (100.0 - 150.0) == pytest.approx(num / den, rel=1e-9)
    assert abs(num / den) < 1.0
    xr = orc.rk4(orc.MODEL_FKIN6, x, u, *tab, 0.05, 1, integrator=orc.INTEG_IRK_RADAU4)
    assert abs((xr[6] - 150.0) / (100.0 - 150.0)) < 0.05 < abs(num / den)
    # RK4 x 1 on the same lag explodes (2.4e5 per step): the reason ERK needs M >= 18
    xe = orc.rk4(orc.MODEL_FKIN6, x, u, *tab, 0.05, 1)
    assert abs((xe[6] - 150.0) / (100.0 - 150.0)) > 1e5


@pytest.mark.parametrize("integ,M", [(orc.INTEG_IRK_GL4, 1), (orc.INTEG_IRK_RADAU4, 3)])
@pytest.mark.parametrize("model", [orc.MODEL_FKIN6, orc.MODEL_FDYN6U])
def test_irk_sensitivities_match_central_differences(tab, model, integ, M):
    rng = np.random.default_rng(11)
    x, u = random_state(rng); x[3] = 8.0
    xn, A, B = orc.rk4_sens(model, x, u, *tab, 0.05, M, integrator=integ)
    np.testing.assert_allclose(xn, orc.rk4(model, x, u, *tab, 0.05, M, integrator=integ), rtol=0, atol=1e-13)
    w = np.concatenate([x, u])
    S = np.hstack([A, B])
    for c in range(10):
        h = 1e-6 * max(1.0, abs(w[c]))
        wp, wm = w.copy(), w.copy(); wp[c] += h; wm[c] -= h
        fd = (orc.rk4(model, wp[:8], wp[8:], *tab, 0.05, M, integrator=integ) - orc.rk4(model, wm[:8], wm[8:], *tab, 0.05, M, integrator=integ)) / (2 * h)
        scale = max(1e-3, np.max(np.abs(fd)))
        # tolerance 5e-4: the sensitivities are those of the collocation equations at the final stage values (implicit-function
        # theorem, as acados), the difference quotient differentiates three Newton iterations from K = 0 -- they differ by the
        # Newton remainder (up to 1.2e-4 on the dynamic model with one step per interval)
        assert np.max(np.abs(S[:, c] - fd)) / scale < 5e-4, (c, S[:, c], fd)


def test_irk_and_fine_rk4_linearisations_agree(tab):
    """A, B of one shooting interval: Radau IIA x 50 against RK4 x 400 (both converged discretisations of the same flow)."""
    rng = np.random.default_rng(3)
    x, u = random_state(rng); x[3] = 7.0
    _, A1, B1 = orc.rk4_sens(orc.MODEL_FKIN6, x, u, *tab, 0.05, 50, integrator=orc.INTEG_IRK_RADAU4)
    _, A2, B2 = orc.rk4_sens(orc.MODEL_FKIN6, x, u, *tab, 0.05, 400)
    assert np.max(np.abs(A1 - A2)) < 1e-8 and np.max(np.abs(B1 - B2)) < 1e-8
