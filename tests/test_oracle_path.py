"""Oracle: the nonlinear track-boundary rows h (old/generate_acaods_interface.py:191-212) in the RTI QP --
values and gradients against the formulas / finite differences, equivalence with the n box in the degenerate
case, and convergence of repeated iterations to a point that respects the rows."""
import numpy as np
from conftest import make_ocp, sample_x0
from test_oracle_rti import _stanley_guess

from oracle import oracle as orc

N = 40
NC = orc.NC


def _path_ocp(soft=False, **kw):
    ocp = make_ocp(**kw)
    ocp.model.con_h_expr = "track"
    c = ocp.constraints
    c.lh = c.lh_e = np.array([-1e3, -1e3])          # the reference's stand-in for "no lower bound"
    c.uh = c.uh_e = np.array([0.0, 0.0])
    if soft:                                          # old/generate_acaods_interface.py:380-395: L1 + L2 weights 100 / 100
        c.idxsh, c.idxsh_e = np.arange(2), np.arange(2)
        ocp.cost.zl = ocp.cost.zu = ocp.cost.Zl = ocp.cost.Zu = np.full(2, 100.0)
        ocp.cost.zl_e = ocp.cost.zu_e = ocp.cost.Zl_e = ocp.cost.Zu_e = np.full(2, 100.0)
    return ocp


def _h(x, L, W, w):
    n, psi = x[..., 1], x[..., 2]
    foot, lat = -0.5 * L * np.sin(np.abs(psi)), 0.5 * W * np.cos(psi)
    return np.stack([n + foot + lat - w[0], -n - foot + lat - w[1]], -1)


def test_track_rows_values_and_gradients(track):
    ocp = _path_ocp()
    d = ocp.flatten()
    w = np.array([[1.9, 1.7]])
    P = orc.OracleProblem(d.as_dict(track.s_ref, track.kappa_ref, track_widths=w))
    rng = np.random.default_rng(3)
    x = rng.normal(size=(N + 1, 8)) * 0.3; x[:, 3] = 8.0; x[:, 0] = np.linspace(5, 40, N + 1)
    x[7, 2] = 0.0                                       # sign(0) = 0 at the kink of |psi|
    u = np.zeros((N, 2))
    qp = P.build_qp(x, u, x[0], np.zeros((N, 12)), np.zeros(8))
    hv = _h(x, d.car_L, d.car_W, w[0])
    assert np.all(np.isinf(qp["dl"][0, 12:])) and np.all(np.isinf(qp["du"][0, 12:])) and np.all(qp["R"][0, 12:] == 0)   # x_0 is fixed
    np.testing.assert_allclose(qp["du"][1:, 12:14], 0.0 - hv[1:], rtol=0, atol=1e-15)
    np.testing.assert_allclose(qp["dl"][1:, 12:14], -1e3 - hv[1:], rtol=0, atol=1e-12)
    eps = 1e-7
    for k in (1, 7, 20, N):
        for j in range(8):
            xp, xm = x[k].copy(), x[k].copy(); xp[j] += eps; xm[j] -= eps
            fd = (_h(xp, d.car_L, d.car_W, w[0]) - _h(xm, d.car_L, d.car_W, w[0])) / (2 * eps)
            if k == 7 and j == 2:
                fd = np.array([-0.0, -0.0])            # symmetric difference across the kink: only the smooth part...
                fd = (-0.5 * d.car_W * np.sin(0.0)) * np.ones(2)
            np.testing.assert_allclose(qp["R"][k, 12:14, j], fd, rtol=0, atol=1e-7)
        assert np.all(qp["R"][k, 12:14, 8:] == 0)


def test_degenerate_rows_equal_the_n_box(track):
    """car_length = car_width = 0 and w_R = w_L = n_max: h_R <= 0 and h_L <= 0 are exactly |n| <= n_max."""
    n_max = 0.6
    B = 5
    x0 = sample_x0(track, B, seed=11); x0[:, 1] = np.linspace(-0.5, 0.5, B)
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(N)[None] / N
    yref[:, :, 1] = 0.9                                   # pull the car towards the bound so that it becomes active
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0; yref_e[:, 1] = 0.9
    outs = []
    for variant in ("box", "rows"):
        # solved tightly: at the default tolerance two different barrier paths (a two-sided box starts its slacks at a quarter of its
        # width, one-sided rows at qp_tau0) stop at different points inside it
        ocp = make_ocp(n_max=n_max, qp_tol=1e-10, qp_solver_iter_max=60) if variant == "box" else _path_ocp(n_max=1e30, qp_tol=1e-10, qp_solver_iter_max=60)
        ocp.constraints.idxbx_e, ocp.constraints.lbx_e, ocp.constraints.ubx_e = (ocp.constraints.idxbx.copy(), ocp.constraints.lbx.copy(),
                                                                                 ocp.constraints.ubx.copy())
        if variant == "rows":
            ocp.model.car_length = ocp.model.car_width = 0.0
            ocp.constraints.lh = ocp.constraints.lh_e = np.array([-1e30, -1e30])
        d = ocp.flatten()
        P = orc.OracleProblem(d.as_dict(track.s_ref, track.kappa_ref, track_widths=[[n_max, n_max]]))
        x, u = _stanley_guess(P, track, x0)
        out = P.rti_step(x, u, x0, yref, yref_e)
        assert np.all(out["status"] == 0)
        outs.append((x, u, out))
    (xb, ub, ob), (xr, ur, orr) = outs
    assert np.abs(xb[:, 1:, 1]).max() > n_max - 1e-4      # the bound is active somewhere
    # two barrier paths to the same QP solution: they meet to the accuracy the (gradient-relative) tolerance gives the weakly weighted
    # states (v_y: weight 1 against gradients of 1e4), 1.3e-4 at qp_tol = 1e-10
    assert np.max(np.abs(xb - xr) / np.maximum(1.0, np.abs(xr))) < 1e-3 and np.max(np.abs(ub - ur) / np.maximum(1.0, np.abs(ur))) < 1e-3
    # the multipliers move from the box columns (1 lower, NC + 1 upper) to the row columns (NC + 12 upper of h_R, NC + 13 of h_L)
    scale = 1.0 + np.abs(ob["lam"]).max()
    assert np.max(np.abs(ob["lam"][:, :, NC + 1] - orr["lam"][:, :, NC + 12])) / scale < 1e-4
    assert np.max(np.abs(ob["lam"][:, :, 1] - orr["lam"][:, :, NC + 13])) / scale < 1e-4


def test_sqp_respects_the_track_rows(track):
    """A narrow track: repeated iterations end at iterates with h <= 0 (hard rows) up to the linearisation defect."""
    ocp = _path_ocp(qp_tol=1e-8, qp_solver_iter_max=60)
    d = ocp.flatten()
    w = np.array([[1.2, 1.2]])
    P = orc.OracleProblem(d.as_dict(track.s_ref, track.kappa_ref, track_widths=w))
    B = 4
    x0 = sample_x0(track, B, seed=5)
    x0[:, 1] = [0.2, -0.2, 0.3, -0.3]; x0[:, 2] *= 0.3; x0[:, 3] = 10.0; x0[:, 6] = 100.0
    x0[:, 5] = x0[:, 3] * np.interp(x0[:, 0], track.s_ref, track.kappa_ref)
    x, u = _stanley_guess(P, track, x0)
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(N)[None] / N
    yref[:, :, 1] = 1.0                                   # reference outside the track
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0; yref_e[:, 1] = 1.0
    pi = lam = None
    for it in range(10):
        out = P.rti_step(x, u, x0, yref, yref_e, pi=pi, lam=lam)
        pi, lam = out["pi"], out["lam"]
        assert np.all(out["status"] == 0)
    hv = _h(x, d.car_L, d.car_W, w[0])[:, 1:]
    assert hv.max() < 1e-6
    assert hv.max() > -1e-3                               # and the right row is active: n + W/2 cos(psi) - ... = w_R
    assert np.abs(lam[:, :, NC + 12]).max() > 1.0
