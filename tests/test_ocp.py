"""OCP description against python/mpc.py:29-101 and python/main.py:193-295."""
import numpy as np
import pytest
from conftest import make_ocp

from ihm2_amd import ocp as O


def test_cost_selectors():
    ocp = make_ocp()
    Vx, Vu = ocp.cost.Vx, ocp.cost.Vu
    nz = {(i, i) for i in range(8)} | {(10, 6), (11, 7)}
    assert {tuple(ix) for ix in np.argwhere(Vx != 0)} == nz and np.all(Vx[Vx != 0] == 1)
    assert Vu[8, 0] == 1 and Vu[9, 1] == 1 and Vu[10, 0] == -1 and Vu[11, 1] == -1 and np.count_nonzero(Vu) == 4
    assert (ocp.dims.ny, ocp.dims.ny_e) == (12, 8)


def test_bounds_and_rate_rows():
    ocp = make_ocp(n_max=0.9)
    c = ocp.constraints
    assert list(c.idxbx) == [1, 3, 6, 7] and list(c.idxbx_e) == [1, 3, 4, 5]     # quirk Q1 reproduced
    np.testing.assert_array_equal(c.lbx, [-0.9, 0.0, -500.0, -0.5])
    np.testing.assert_array_equal(c.ubx, [0.9, 31.0, 500.0, 0.5])
    np.testing.assert_array_equal(c.lbx_e, [-0.9, -31.0, -500.0, -0.5])
    np.testing.assert_array_equal(c.lbu, [-500.0, -0.5])
    np.testing.assert_allclose(c.ug, [1e-3 * 1e6, 0.02 * 1.0])
    np.testing.assert_allclose(c.lg, [-1000.0, -0.02])
    assert c.C[0, 6] == -1 and c.C[1, 7] == -1 and np.count_nonzero(c.C) == 2
    np.testing.assert_array_equal(c.D, np.eye(2))


def test_flatten_per_stage_arrays():
    d = make_ocp(N=20).flatten()
    assert d.N == 20 and d.dt == pytest.approx(0.05) and d.cost_scale_stage == pytest.approx(0.05)
    assert d.W.shape == (20, 12, 12) and d.lbx.shape == (21, 8) and d.C.shape == (20, 2, 8)
    assert np.all(np.isinf(d.lbx[0])) and np.all(np.isinf(d.ubx[0]))          # x_0 is fixed, not boxed
    assert np.isfinite(d.lbx[1]).tolist() == [False, True, False, True, False, False, True, True]
    assert np.isfinite(d.lbx[20]).tolist() == [False, True, False, True, True, True, False, False]
    np.testing.assert_array_equal(np.diag(d.W[0]), [1, 1, 1, 1, 1, 1, 1, 100, 1, 100, 0, 500])
    np.testing.assert_array_equal(np.diag(d.W_e), [1000, 100, 100, 1, 1, 1, 1, 100])
    # constant Gauss-Newton Hessian block: (delta, u_delta) 2x2 = [[600,-500],[-500,600]]
    V = np.hstack([make_ocp().cost.Vx, make_ocp().cost.Vu])
    H = V.T @ d.W[0] @ V
    np.testing.assert_array_equal(H[np.ix_([7, 9], [7, 9])], [[600, -500], [-500, 600]])
    assert np.linalg.eigvalsh(H).min() == pytest.approx(1.0)


def test_unsupported_options_are_rejected():
    ocp = make_ocp()
    ocp.solver_options.integrator_type = "IRK"
    with pytest.raises(ValueError):
        ocp.flatten()
    ocp = make_ocp()
    ocp.cost.Vx = np.zeros((12, 8))
    with pytest.raises(ValueError):
        ocp.flatten()
    with pytest.raises(ValueError):
        O.get_acados_model_from_explicit_dynamics("x", "kin4", 8, 2, 10)
