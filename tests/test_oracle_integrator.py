"""Pins the oracle's integrator: RK4 x M against a high-accuracy stiff reference (scipy Radau),
order of convergence, the stiffness regression of SURVEY.md F4, and sensitivities against
finite differences of the discrete map."""
import numpy as np
import pytest
from conftest import random_state
from scipy.integrate import solve_ivp

from oracle import models_np as mnp
from oracle import oracle as orc

DT = 0.05


def _reference_flow(fnp, x, u, track):
    sol = solve_ivp(lambda t, y: fnp(y, u, track.s_ref, track.kappa_ref), (0.0, DT), x, method="Radau",
                    rtol=1e-12, atol=1e-13)
    return sol.y[:, -1]


@pytest.mark.parametrize("model,fnp", [(orc.MODEL_FKIN6, mnp.fkin6), (orc.MODEL_FDYN6, mnp.fdyn6)])
def test_rk4_converges_to_stiff_reference(track, model, fnp):
    rng = np.random.default_rng(10)
    x, u = random_state(rng)
    x[3] = 8.0
    ref = _reference_flow(fnp, x, u, track)
    errs = []
    for M in (50, 100, 200):
        xn = orc.rk4(model, x, u, track.s_ref, track.kappa_ref, DT, M)
        errs.append(np.max(np.abs(xn - ref) / (1 + np.abs(ref))))
    assert errs[0] < 1e-5
    assert errs[2] < 1e-8
    # 4th order: halving h divides the error by ~16 (allow slack for the reference's own error)
    assert errs[0] / errs[1] > 8.0


def test_stiffness_regression_F4(track):
    """t_T = 1e-3, dt = 0.05: one RK4 step amplifies the torque error by R(-50) = 2.4e5;
    M = 25 brings it to R(-2)^25 = 1.2e-12."""
    x = np.array([10.0, 0.0, 0.0, 5.0, 0.0, 0.0, 100.0, 0.0])
    u = np.array([0.0, 0.0])
    _, A1, _ = orc.rk4_sens(orc.MODEL_FKIN6, x, u, track.s_ref, track.kappa_ref, DT, 1)
    z = -50.0
    assert A1[6, 6] == pytest.approx(1 + z + z**2 / 2 + z**3 / 6 + z**4 / 24, rel=1e-12)
    assert abs(A1[6, 6]) > 2e5
    _, A25, B25 = orc.rk4_sens(orc.MODEL_FKIN6, x, u, track.s_ref, track.kappa_ref, DT, 25)
    z = -2.0
    R = 1 + z + z**2 / 2 + z**3 / 6 + z**4 / 24
    assert A25[6, 6] == pytest.approx(R**25, rel=1e-9)
    assert B25[6, 0] == pytest.approx(1 - R**25, rel=1e-12)
    # steering lag, z = -2.5 per interval: exact e^-2.5 = 0.082; M=25 reproduces it
    assert A25[7, 7] == pytest.approx(np.exp(-2.5), rel=1e-5)


@pytest.mark.parametrize("model", [orc.MODEL_FKIN6, orc.MODEL_FDYN6])
@pytest.mark.parametrize("M", [20, 40])
def test_sensitivities_match_finite_differences(track, model, M):
    rng = np.random.default_rng(11)
    for _ in range(3):
        x, u = random_state(rng)
        x[3] = rng.uniform(4, 15)
        # keep s away from a knot so the discrete map is smooth around x
        xn, A, Bm = orc.rk4_sens(model, x, u, track.s_ref, track.kappa_ref, DT, M)
        S = np.hstack([A, Bm])
        Sfd = np.zeros((8, 10))
        for j in range(10):
            h = (1e-5, 1e-5, 1e-5, 1e-5, 1e-5, 1e-5, 1e-1, 1e-5, 1e-1, 1e-5)[j]
            xp, up, xm, um = x.copy(), u.copy(), x.copy(), u.copy()
            if j < 8:
                xp[j] += h; xm[j] -= h
            else:
                up[j - 8] += h; um[j - 8] -= h
            fp = orc.rk4(model, xp, up, track.s_ref, track.kappa_ref, DT, M)
            fm = orc.rk4(model, xm, um, track.s_ref, track.kappa_ref, DT, M)
            Sfd[:, j] = (fp - fm) / (2 * h)
        # column-scaled comparison (T-columns are ~1e-3 of the others)
        err = np.abs(S - Sfd) / (1e-6 + np.abs(Sfd).max(axis=0, keepdims=True))
        assert err.max() < 1e-5, err.max()
        # rk4_sens returns the same x_next as rk4
        np.testing.assert_allclose(xn, orc.rk4(model, x, u, track.s_ref, track.kappa_ref, DT, M), rtol=1e-13, atol=1e-14)


def test_block_triangular_structure_of_A(track):
    """(T,delta) -> (v_x,v_y,r) -> (s,n,psi): SURVEY.md Appendix C.1."""
    x, u = random_state(np.random.default_rng(12))
    _, A, Bm = orc.rk4_sens(orc.MODEL_FKIN6, x, u, track.s_ref, track.kappa_ref, DT, 25)
    assert np.all(A[3:, :3] == 0)
    assert np.all(A[6:, :6] == 0)
    assert A[6, 7] == 0 and A[7, 6] == 0
    assert np.all(Bm[6:, :] == np.diag(np.diag(Bm[6:, :])))
