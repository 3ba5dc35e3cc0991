"""SURVEY.md section 8f N2 remainder: ``fdyn10`` (python/models.py:609-801), the 15-state Frenet plant with wheel speeds.  The C oracle's
explicit form against the reference's IMPLICIT residual restated independently in NumPy (oracle/models_np.py), physical sanity, RK4."""
import numpy as np
import pytest


def _states(track, n, seed=3):
    from oracle import models_np as M

    rng = np.random.default_rng(seed)
    x = np.zeros((n, 15)); u = np.zeros((n, 5))
    x[:, 0] = rng.uniform(0, track.lap_length, n); x[:, 1] = rng.uniform(-0.5, 0.5, n); x[:, 2] = rng.uniform(-0.1, 0.1, n)
    x[:, 3] = rng.uniform(3, 15, n); x[:, 4] = rng.uniform(-0.3, 0.3, n); x[:, 5] = rng.uniform(-0.4, 0.4, n)
    x[:, 6:10] = x[:, 3:4] / M.R_w * (1.0 + rng.uniform(-0.03, 0.06, (n, 4)))        # wheel speeds around rolling
    x[:, 10:14] = rng.uniform(-20, 60, (n, 4)); x[:, 14] = rng.uniform(-0.2, 0.2, n)
    u[:, :4] = rng.uniform(-20, 60, (n, 4)); u[:, 4] = rng.uniform(-0.2, 0.2, n)
    return x, u


def test_explicit_form_solves_the_reference_implicit_residual(track):
    from oracle import models_np as M, oracle as orc

    x, u = _states(track, 200)
    xd = orc.f_dyn10(x, u, track.s_ref, track.kappa_ref)
    assert np.all(np.isfinite(xd))
    for b in range(x.shape[0]):
        res = M.fdyn10_residual(xd[b], x[b], u[b], track.s_ref, track.kappa_ref)
        scale = np.maximum(1.0, np.abs(np.array([1, 1, 1, M.m * xd[b, 3], M.m * xd[b, 4], M.I_z * xd[b, 5]] + [M.I_w * v for v in xd[b, 6:10]] + [1] * 5)))
        assert np.max(np.abs(res) / scale) < 1e-11, (b, res)


def test_straight_rolling_car_is_symmetric_and_decelerates_by_drag(track):
    from oracle import models_np as M, oracle as orc

    # straight piece of a synthetic track (kappa = 0), no steering, no torque: left = right, no lateral motion, v_x falls
    s_ref = np.linspace(-10.0, 500.0, 50); kappa = np.zeros(50)
    x = np.zeros((1, 15)); x[0, 3] = 10.0; x[0, 6:10] = 10.0 / M.R_w
    xd = orc.f_dyn10(x, np.zeros((1, 5)), s_ref, kappa)[0]
    assert abs(xd[4]) < 1e-12 and abs(xd[5]) < 1e-12 and abs(xd[1]) < 1e-12 and abs(xd[2]) < 1e-12
    assert xd[0] == pytest.approx(10.0) and xd[3] < 0.0
    assert xd[6] == pytest.approx(xd[7]) and xd[8] == pytest.approx(xd[9])
    # drive torque on all wheels: the car accelerates, the wheels spin up first
    x[0, 10:14] = 40.0
    xd = orc.f_dyn10(x, np.full((1, 5), 40.0) * np.array([1, 1, 1, 1, 0]), s_ref, kappa)[0]
    assert np.all(xd[6:10] > 0.0)


def test_rk4_plant_step_converges_with_fourth_order(track):
    from oracle import oracle as orc

    x, u = _states(track, 8, seed=9)
    ref = orc.sim_step_dyn10(x, u, track.s_ref, track.kappa_ref, 3200, dt=0.01)
    e = [np.max(np.abs(orc.sim_step_dyn10(x, u, track.s_ref, track.kappa_ref, M, dt=0.01) - ref)) for M in (100, 200)]
    assert e[0] < 1e-6 and e[1] < e[0] / 10.0          # ~ 2^-4 per halving of the step


# ---- the reference's plant integrator on this model (python/main.py:395-400: IRK, GAUSS_RADAU_IIA, 4 stages, 100 steps), from rest ----
_RADAU_A = np.array([[0.11299947932315618599, -0.040309220723522205736, 0.025802377420336391036, -0.0099046765072664238987],
                     [0.23438399574740025657, 0.20689257393535890010, -0.047857128048540718850, 0.016047422806516273037],
                     [0.21668178462325034184, 0.40612326386737331123, 0.18903651817005634247, -0.024182104899832939517],
                     [0.22046221117676837528, 0.38819346884317188078, 0.32884431998005974394, 0.0625]])


def _rest():
    x0 = np.zeros(15); x0[0] = -6.0           # python/main.py:438-441
    return x0


def test_dual_number_jacobian_of_fdyn10_matches_central_differences(track):
    from oracle import oracle as orc

    x, u = _states(track, 6, seed=3)
    for b in range(6):
        f, J = orc.jac_dyn10(x[b], u[b], track.s_ref, track.kappa_ref)
        np.testing.assert_allclose(f, orc.f_dyn10(x[b], u[b], track.s_ref, track.kappa_ref)[0], rtol=1e-12, atol=1e-9)
        Jfd = np.zeros((15, 15))
        for j in range(15):
            h = 1e-6 * (1 + abs(x[b, j])); xp = x[b].copy(); xp[j] += h; xm = x[b].copy(); xm[j] -= h
            Jfd[:, j] = (orc.f_dyn10(xp, u[b], track.s_ref, track.kappa_ref)[0] - orc.f_dyn10(xm, u[b], track.s_ref, track.kappa_ref)[0]) / (2 * h)
        assert np.max(np.abs(J - Jfd)) <= 1e-7 * np.max(np.abs(J))


def test_a_fixed_number_of_newton_iterations_does_not_converge_at_rest(track):
    """acados' IRK runs newton_iter = 3 iterations per step from K = 0.  At v = 0 the rolling condition of a wheel has a basin of
    1e-7 rad/s (slip ratio omega R_w / 1e-6 - 1): 3 and 6 iterations give different states after ONE plant step, neither near the
    solution -- which is why the plant of this build iterates to convergence and lets the step length follow the iteration."""
    from oracle import oracle as orc

    u = np.array([25.0, 25.0, 25.0, 25.0, 0.0])

    def fixed_newton(n_iter, M=100, dt=0.05):
        x, h = _rest(), dt / M
        for _ in range(M):
            K = np.zeros((4, 15))
            for _ in range(n_iter):
                Mx = np.eye(60); R = np.zeros((4, 15))
                for i in range(4):
                    f, J = orc.jac_dyn10(x + h * (_RADAU_A[i] @ K), u, track.s_ref, track.kappa_ref)
                    R[i] = K[i] - f
                    for j in range(4):
                        Mx[i * 15:(i + 1) * 15, j * 15:(j + 1) * 15] -= h * _RADAU_A[i, j] * J
                K = K + np.linalg.solve(Mx, -R.ravel()).reshape(4, 15)
            x = x + h * (_RADAU_A[3] @ K)
        return x

    x3, x6 = fixed_newton(3), fixed_newton(6)
    xo = orc.sim_step_dyn10_irk(_rest(), u, track.s_ref, track.kappa_ref)[0]
    assert np.all(np.isfinite(xo)) and abs(xo[9]) < 1e-3            # 25 N m per wheel do not overcome the rolling resistance: the car creeps
    assert abs(x3[9] - xo[9]) > 0.1 and abs(x6[9] - xo[9]) > 0.1 and abs(x3[9] - x6[9]) > 0.1       # rad/s on a wheel that stands still


@pytest.mark.parametrize("u_tau,u_delta", [(25.0, 0.0), (125.0, 0.05), (60.0, -0.1)])
def test_radau_plant_from_rest_matches_scipy_radau(track, u_tau, u_delta):
    """x0 = (-6, 0, ..., 0) as python/main.py:438-441, three plant steps of 0.05 s: the oracle's Radau IIA x 100 (solved to convergence,
    step cuts where the iteration asks for them) against scipy's adaptive Radau at rtol 1e-12 -- tolerance 1e-9 relative."""
    from scipy.integrate import solve_ivp

    from oracle import oracle as orc

    u = np.array([u_tau] * 4 + [u_delta])
    fun = lambda t, y: orc.f_dyn10(y, u, track.s_ref, track.kappa_ref)[0]
    jac = lambda t, y: orc.jac_dyn10(y, u, track.s_ref, track.kappa_ref)[1]
    xo = xs = _rest()
    for _ in range(3):
        xo = orc.sim_step_dyn10_irk(xo, u, track.s_ref, track.kappa_ref)[0]
        xs = solve_ivp(fun, (0, 0.05), xs, method="Radau", jac=jac, rtol=1e-12, atol=1e-14).y[:, -1]
        assert np.all(np.isfinite(xo))
        assert np.max(np.abs(xo - xs) / (1.0 + np.abs(xs))) < 1e-9
    if u_tau > 100:
        assert xo[3] > 0.9          # the car is under way (1 m/s after 0.15 s at full torque)


def test_radau_and_rk4_agree_on_a_moving_car(track):
    from oracle import oracle as orc

    x, u = _states(track, 12, seed=8)
    xr = orc.sim_step_dyn10_irk(x, u, track.s_ref, track.kappa_ref)
    xk = orc.sim_step_dyn10(x, u, track.s_ref, track.kappa_ref, 400, dt=0.05)
    assert np.all(np.isfinite(xr)) and np.max(np.abs(xr - xk) / (1.0 + np.abs(xk))) < 1e-8
