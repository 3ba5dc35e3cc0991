"""Pins the C oracle's vehicle models: against an independent NumPy restatement of
python/models.py (oracle/models_np.py), by complex-step Jacobians, and by closed-form spot values."""
import json
import os

import numpy as np
import pytest
from conftest import random_state

from oracle import models_np as mnp
from oracle import oracle as orc

MODELS = [(orc.MODEL_FKIN6, mnp.fkin6), (orc.MODEL_FDYN6, mnp.fdyn6), (orc.MODEL_FDYN6U, mnp.fdyn6u)]


def test_numpy_mirror_constants_match_golden():
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "constants.json")))["values"]
    for name in ("m", "I_z", "l_R", "l_F", "wheelbase", "axle_track", "z_CG", "C_m0", "C_r0", "C_r1", "C_r2",
                 "Ba", "Ca", "Da", "Ea", "t_T", "t_delta", "C_downforce", "K_tv", "static_weight"):
        assert getattr(mnp, name) == pytest.approx(gold[name], rel=1e-15), name


@pytest.mark.parametrize("model,fnp", MODELS)
def test_f_matches_numpy_mirror(track, model, fnp):
    rng = np.random.default_rng(1)
    for _ in range(300):
        x, u = random_state(rng)
        fc = orc.f(model, x, u, track.s_ref, track.kappa_ref)
        fn = fnp(x, u, track.s_ref, track.kappa_ref)
        np.testing.assert_allclose(fc, fn, rtol=1e-12, atol=1e-11)


@pytest.mark.parametrize("model,fnp", MODELS)
def test_jacobian_matches_complex_step(track, model, fnp):
    rng = np.random.default_rng(2)
    for _ in range(100):
        x, u = random_state(rng)
        _, J = orc.jac(model, x, u, track.s_ref, track.kappa_ref)
        _, Jcs = orc.jac(model, x, u, track.s_ref, track.kappa_ref, complex_step=True)
        Jn = mnp.jac_complex_step(fnp, x, u, track.s_ref, track.kappa_ref)
        scale = 1.0 + np.abs(Jn)
        assert np.max(np.abs(J - Jn) / scale) < 1e-11
        assert np.max(np.abs(Jcs - Jn) / scale) < 1e-11


def test_fkin6_jacobian_sparsity(track):
    """31 structural non-zeros."""
    rng = np.random.default_rng(3)
    pattern = np.zeros((8, 10), dtype=bool)
    for _ in range(20):
        x, u = random_state(rng)
        _, J = orc.jac(orc.MODEL_FKIN6, x, u, track.s_ref, track.kappa_ref)
        pattern |= J != 0
    assert pattern.sum() == 31
    assert not pattern[3:, :3].any()          # s, n, psi never enter rows 4-8
    assert pattern[:, 5].sum() == 1            # r enters psi_dot only


def test_fkin6_rest_state_is_equilibrium(track):
    """x = 0, u = 0 => xdot = 0 since tanh(0) = 0 (SURVEY.md 8c check 1)."""
    f = orc.f(orc.MODEL_FKIN6, np.zeros(8), np.zeros(2), track.s_ref, track.kappa_ref)
    assert np.all(f == 0.0)


def test_actuator_rows(track):
    x, u = random_state(np.random.default_rng(4))
    for model, _ in MODELS:
        f = orc.f(model, x, u, track.s_ref, track.kappa_ref)
        assert f[6] == pytest.approx((u[0] - x[6]) / 1e-3, rel=1e-14)
        assert f[7] == pytest.approx((u[1] - x[7]) / 0.02, rel=1e-14)


def test_fdyn6_explicit_solves_reference_implicit_residual(track):
    rng = np.random.default_rng(5)
    for _ in range(50):
        x, u = random_state(rng)
        xdot = orc.f(orc.MODEL_FDYN6, x, u, track.s_ref, track.kappa_ref)
        res = mnp.fdyn6_residual(xdot, x, u, track.s_ref, track.kappa_ref)
        scale = np.array([1, 1, 1, mnp.m * 10, mnp.m * 10, mnp.I_z * 10, 1e3, 1e2])
        assert np.max(np.abs(res) / scale) < 1e-11


def test_fdyn6_as_written_is_open_loop_unstable_and_uncrossed_is_not(track):
    """Quirk Q3 (python/models.py:543-546: F_lat_FL uses alpha_RR, ...): the crossed slip angles turn the yaw/sideslip
    feedback positive.  Straight driving at 10 m/s: eigenvalue +34 1/s as written, all <= 1 1/s un-crossed (the +0.84 is the
    Frenet kinematics n' = v psi, shared with fkin6)."""
    x = np.array([50.0, 0.0, 0.0, 10.0, 0.0, 0.0, 50.0, 0.0]); u = np.array([50.0, 0.0])
    ev = {}
    for mdl in (orc.MODEL_FKIN6, orc.MODEL_FDYN6, orc.MODEL_FDYN6U):
        _, J = orc.jac(mdl, x, u, track.s_ref, track.kappa_ref)
        ev[mdl] = np.linalg.eigvals(J[:, :8]).real.max()
    assert ev[orc.MODEL_FDYN6] > 30.0
    assert ev[orc.MODEL_FDYN6U] < 1.0 and abs(ev[orc.MODEL_FDYN6U] - ev[orc.MODEL_FKIN6]) < 1e-3
    # and over one shooting interval (dt = 0.05, RK4 x 25): spectral radius 5.6 against 1.04
    for mdl, lo, hi in ((orc.MODEL_FDYN6, 5.0, 6.5), (orc.MODEL_FDYN6U, 1.0, 1.1)):
        _, A, _ = orc.rk4_sens(mdl, x, u, track.s_ref, track.kappa_ref, 0.05, 25)
        assert lo < np.abs(np.linalg.eigvals(A)).max() < hi


def test_kappa_interpolant(track):
    s_ref, k_ref = track.s_ref, track.kappa_ref
    rng = np.random.default_rng(6)
    for s in rng.uniform(s_ref[0], s_ref[-1], 200):
        v, dk = orc.kappa(s_ref, k_ref, s)
        assert v == pytest.approx(np.interp(s, s_ref, k_ref), rel=1e-12, abs=1e-15)
    # knots are reproduced and the slope is that of the segment to the right
    for i in (0, 7, 499, 500, 1498):
        v, dk = orc.kappa(s_ref, k_ref, s_ref[i])
        assert v == pytest.approx(k_ref[i], abs=1e-15)
        assert dk == pytest.approx((k_ref[i + 1] - k_ref[i]) / (s_ref[i + 1] - s_ref[i]), rel=1e-12)
    # linear extrapolation outside the grid (casadi linear interpolant)
    v, dk = orc.kappa(s_ref, k_ref, s_ref[-1] + 3.0)
    slope = (k_ref[-1] - k_ref[-2]) / (s_ref[-1] - s_ref[-2])
    assert v == pytest.approx(k_ref[-1] + 3.0 * slope, rel=1e-12)
    v, dk = orc.kappa(s_ref, k_ref, s_ref[0] - 2.0)
    slope = (k_ref[1] - k_ref[0]) / (s_ref[1] - s_ref[0])
    assert v == pytest.approx(k_ref[0] - 2.0 * slope, rel=1e-12)
