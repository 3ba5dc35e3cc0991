"""Oracle: the lateral-acceleration row of the kinematic constraint set -- old/generate_acaods_interface.py:198-209
(`[right, left, T_dot, delta_dot] + ([] if is_dynamic else [a_lat])`), definition :266-271 / old/scripts/gen_mpc.py:182-184,
bounds :52-53 (-5 / +5 m/s^2).  Value and gradient against the formula written out with numpy and finite differences, the
kinematic identity a_lat = v * yaw rate without forces, the fifteen-row build against the fourteen-row one when the row is
idle, and repeated iterations that end at iterates respecting it."""
import numpy as np
import pytest
from conftest import make_ocp, sample_x0
from test_oracle_rti import _stanley_guess

from ihm2_amd import constants as K
from oracle import oracle as orc

N = 40
NC = orc.NC


def _alat_np(x):
    """The reference's expression, term by term (tan / atan / sin as it writes them), on the live kinematic model's forces."""
    v_x, v_y, T, delta = x[..., 3], x[..., 4], x[..., 6], x[..., 7]
    F_motor = K.C_m0 * T
    F_drag = -(K.C_r0 + K.C_r1 * v_x + K.C_r2 * v_x * v_x) * np.tanh(10.0 * v_x)
    F_Rx, F_Fx = 0.5 * F_motor + F_drag, 0.5 * F_motor
    C = K.l_R / (K.l_R + K.l_F)
    beta = np.arctan(C * np.tan(delta))
    return (-F_Rx * np.sin(beta) + F_Fx * np.sin(delta - beta)) / K.m + (v_x * v_x + v_y * v_y) * np.sin(beta) / K.l_R


def _alat_ocp(soft=False, a_max=5.0, **kw):
    ocp = make_ocp(**kw)
    ocp.model.con_h_expr = "track+a_lat"
    c = ocp.constraints
    c.lh = np.array([-1e3, -1e3, -a_max]); c.uh = np.array([0.0, 0.0, a_max])       # :418-435
    c.lh_e = np.array([-1e3, -1e3]); c.uh_e = np.array([0.0, 0.0])
    if soft:        # every h row soft, L1 + L2 weights 100 / 100 (:380-395)
        c.idxsh, c.idxsh_e = np.arange(3), np.arange(2)
        ocp.cost.zl = ocp.cost.zu = ocp.cost.Zl = ocp.cost.Zu = np.full(3, 100.0)
        ocp.cost.zl_e = ocp.cost.zu_e = ocp.cost.Zl_e = ocp.cost.Zu_e = np.full(2, 100.0)
    return ocp


def test_value_and_gradient():
    rng = np.random.default_rng(1)
    for _ in range(50):
        x = rng.normal(size=8)
        x[3] = rng.uniform(0.5, 25.0); x[4] = rng.normal() * 0.5; x[6] = rng.uniform(-300, 500); x[7] = rng.uniform(-0.5, 0.5)
        val, g = orc.alat(x)
        assert abs(val - _alat_np(x)) <= 1e-12 * max(1.0, abs(val))
        assert np.all(g[[0, 1, 2, 5]] == 0.0)              # the row's non-zeros: v_x, v_y, T, delta
        for j in (3, 4, 6, 7):
            h = 1e-6 * max(1.0, abs(x[j]))
            xp, xm = x.copy(), x.copy(); xp[j] += h; xm[j] -= h
            fd = (_alat_np(xp) - _alat_np(xm)) / (2 * h)
            assert abs(g[j] - fd) <= 1e-6 * max(1.0, abs(fd)), (j, g[j], fd)


def test_without_forces_it_is_speed_times_yaw_rate():
    """T = 0 and no drag left (v_x -> the forces vanish like v): a_lat -> v^2 sin(beta) / l_R = v * (v sin(beta) / l_R), the kinematic
    model's yaw rate (python/models.py:304: r_dot follows l_R v_y_dot - beta_dot; old: r = v sin(beta) / l_R) times the speed."""
    x = np.zeros(8); x[3] = 12.0; x[7] = 0.2
    C = K.l_R / (K.l_R + K.l_F)
    beta = np.arctan(C * np.tan(0.2))
    val, _ = orc.alat(x)
    drag = (K.C_r0 + K.C_r1 * 12.0 + K.C_r2 * 144.0) * np.tanh(120.0)
    assert abs(val - (144.0 * np.sin(beta) / K.l_R + drag * np.sin(beta) / K.m)) < 1e-12


def test_row_in_the_qp_data(track):
    d = _alat_ocp().flatten()
    assert d.alat_on == 1 and d.alat_lb == -5.0 and d.alat_ub == 5.0 and d.lh.shape == (2,)
    P = orc.OracleProblem(d.as_dict(track.s_ref, track.kappa_ref, track_widths=[[1.9, 1.7]]))
    assert P.nc == 15
    rng = np.random.default_rng(3)
    x = rng.normal(size=(N + 1, 8)) * 0.3; x[:, 3] = np.linspace(4, 14, N + 1); x[:, 0] = np.linspace(5, 40, N + 1); x[:, 6] = 80.0
    u = np.zeros((N, 2))
    qp = P.build_qp(x, u, x[0], np.zeros((N, 12)), np.zeros(8))
    assert qp["R"].shape == (N + 1, 15, 10)
    for k in (0, N):        # x_0 is fixed; no terminal row (con_h_expr_e: the track rows only)
        assert np.isinf(qp["dl"][k, 14]) and np.isinf(qp["du"][k, 14]) and np.all(qp["R"][k, 14] == 0)
    av = _alat_np(x)
    np.testing.assert_allclose(qp["du"][1:N, 14], 5.0 - av[1:N], rtol=0, atol=1e-12)
    np.testing.assert_allclose(qp["dl"][1:N, 14], -5.0 - av[1:N], rtol=0, atol=1e-12)
    for k in (1, 17, N - 1):
        _, g = orc.alat(x[k])
        np.testing.assert_array_equal(qp["R"][k, 14, :8], g)
        assert np.all(qp["R"][k, 14, 8:] == 0)
    # the fourteen rows in front of it are those of the fourteen-row build
    d14 = _alat_ocp().flatten(); d14.alat_on = 0
    qp14 = orc.OracleProblem(d14.as_dict(track.s_ref, track.kappa_ref, track_widths=[[1.9, 1.7]])).build_qp(x, u, x[0], np.zeros((N, 12)), np.zeros(8))
    np.testing.assert_array_equal(qp["R"][:, :14], qp14["R"]); np.testing.assert_array_equal(qp["dl"][:, :14], qp14["dl"])


def _problem(track, B, seed, **kw):
    x0 = sample_x0(track, B, seed=seed)
    x0[:, 3] = np.linspace(9.0, 14.0, B)
    a_start = kw.get("a_start")
    if a_start is not None:        # a hard row needs a start inside it: slow the car down until |a_lat(x0)| <= a_start
        for _ in range(60):
            x0[:, 3] = np.where(np.abs(_alat_np(x0)) > a_start, 0.95 * x0[:, 3], x0[:, 3])
    x0[:, 5] = x0[:, 3] * np.interp(x0[:, 0], track.s_ref, track.kappa_ref)
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 60.0 * np.arange(N)[None] / N; yref[:, :, 3] = 15.0
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 60.0; yref_e[:, 3] = 15.0
    return x0, yref, yref_e


def test_an_idle_row_changes_nothing(track):
    """Bounds the iterates never come near: the fifteen-row QP has the solution of the fourteen-row one (a row far from its bound carries
    barrier weight ~ mu / t^2: the two agree to the QP tolerance) and the row's multipliers vanish."""
    B = 6
    x0, yref, yref_e = _problem(track, B, 21)
    outs = []
    for a_max in (None, 1e4):
        ocp = _alat_ocp(a_max=1e4, qp_tol=1e-10, qp_solver_iter_max=60)
        d = ocp.flatten()
        if a_max is None:
            d.alat_on = 0
        P = orc.OracleProblem(d.as_dict(track.s_ref, track.kappa_ref, track_widths=[[1.6, 1.6]]))
        x, u = _stanley_guess(P, track, x0)
        out = P.rti_step(x, u, x0, yref, yref_e)
        assert np.all(out["status"] == 0)
        outs.append((x, u, out))
    (x14, u14, o14), (x15, u15, o15) = outs
    assert o15["lam"].shape[-1] == 30
    assert np.max(np.abs(x14 - x15) / np.maximum(1.0, np.abs(x14))) < 1e-6 and np.max(np.abs(u14 - u15) / np.maximum(1.0, np.abs(u14))) < 1e-6
    assert np.abs(o15["lam"][:, :, [14, 29]]).max() < 1e-6 * (1.0 + np.abs(o15["lam"]).max())


@pytest.mark.parametrize("soft", [False, True])
def test_iterations_respect_the_row(track, soft):
    """A tight bound (2 m/s^2) on a track driven at 9-14 m/s: the unconstrained iterates exceed it; with the row, repeated iterations end
    at iterates with |a_lat| <= 2 (hard) or within the slack the L1 + L2 penalty buys (soft), with the row active."""
    B = 6
    a_max = 2.0
    x0, yref, yref_e = _problem(track, B, 22, a_start=None if soft else 1.0)
    d = _alat_ocp(soft=soft, a_max=a_max, qp_tol=1e-8, qp_solver_iter_max=80).flatten()
    w = [[1.6, 1.6]]
    P = orc.OracleProblem(d.as_dict(track.s_ref, track.kappa_ref, track_widths=w))
    d0 = _alat_ocp(soft=soft, a_max=a_max, qp_tol=1e-8, qp_solver_iter_max=80).flatten(); d0.alat_on = 0
    if soft:
        assert d.soft_Z.shape == (N + 1, 28) and np.all(d.alat_soft_Z == 100.0)
    P0 = orc.OracleProblem(d0.as_dict(track.s_ref, track.kappa_ref, track_widths=w))
    res = {}
    for name, prob in (("free", P0), ("row", P)):
        x, u = _stanley_guess(prob, track, x0)
        pi = lam = None
        for _ in range(8):
            out = prob.rti_step(x, u, x0, yref, yref_e, pi=pi, lam=lam)
            pi, lam = out["pi"], out["lam"]
        ok = out["status"] == 0
        assert ok.sum() >= B - 1
        res[name] = (np.abs(_alat_np(x[ok]))[:, 1:N], lam[ok])
    assert res["free"][0].max() > a_max + 0.5          # the row matters on this track
    worst = res["row"][0].max()
    if soft:
        # cars that start far outside the row (|a_lat(x0)| up to 10 m/s^2) buy slack at 100 s + 50 s^2: the excess shrinks, it does not vanish
        assert worst < 0.5 * res["free"][0].max()
        excess = lambda a: np.maximum(a - a_max, 0.0).sum()
        assert excess(res["row"][0]) < 0.25 * excess(res["free"][0])
    else:
        assert worst < a_max + 1e-5
    assert worst > a_max - 1e-3                         # active
    assert np.abs(res["row"][1][:, :, [14, 29]]).max() > 1e-2
