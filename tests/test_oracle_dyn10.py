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
