"""GPU: the lateral-acceleration row of the kinematic constraint set (old/generate_acaods_interface.py:198-209; definition :266-271 /
old/scripts/gen_mpc.py:182-184) -- `k_qp_wave<10,4,2,1>` through the C ABI (ihm2mpc_set_alat_constraint) against the fifteen-row build
of the oracle: statuses, iteration counts, iterates, KKT residuals, multipliers and slacks, hard and soft, on small batches step by step
and on 4096 instances in both scheduler builds; an idle row against the kernel without it; the refusals."""
import numpy as np
import pytest
from conftest import make_ocp, sample_x0

pytestmark = pytest.mark.gpu
N = 40


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def _alat_np(x):
    from ihm2_amd import constants as K

    v_x, v_y, T, delta = x[..., 3], x[..., 4], x[..., 6], x[..., 7]
    F_drag = -(K.C_r0 + K.C_r1 * v_x + K.C_r2 * v_x * v_x) * np.tanh(10.0 * v_x)
    F_Rx, F_Fx = 0.5 * K.C_m0 * T + F_drag, 0.5 * K.C_m0 * T
    beta = np.arctan(K.l_R / (K.l_R + K.l_F) * np.tan(delta))
    return (-F_Rx * np.sin(beta) + F_Fx * np.sin(delta - beta)) / K.m + (v_x * v_x + v_y * v_y) * np.sin(beta) / K.l_R


def _setup(track, soft, B, seed, a_max, width=1.6, row=True, penalty=100.0, **opts):
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import oracle as orc

    ocp = make_ocp(**opts)
    ocp.model.con_h_expr = "track+a_lat" if row else "track"
    c = ocp.constraints
    c.lh = np.array([-1e3, -1e3, -a_max][:2 + row]); c.uh = np.array([0.0, 0.0, a_max][:2 + row])
    c.lh_e = np.array([-1e3, -1e3]); c.uh_e = np.array([0.0, 0.0])
    if soft:        # every h row soft, L1 + L2 weights 100 / 100 (old/generate_acaods_interface.py:380-395, idxsh = all)
        c.idxsh, c.idxsh_e = np.arange(2 + row), np.arange(2)
        ocp.cost.zl = ocp.cost.zu = ocp.cost.Zl = ocp.cost.Zu = np.array([100.0, 100.0, penalty][:2 + row])
        ocp.cost.zl_e = ocp.cost.zu_e = ocp.cost.Zl_e = ocp.cost.Zu_e = np.full(2, 100.0)
    w = np.array([[width, width - 0.1]])
    solver = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref, track_widths=w)
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref, track_widths=w))
    x0 = sample_x0(track, B, seed=seed)
    x0[:, 3] = np.linspace(8.0, 14.0, B)
    if not soft:        # a hard row needs a start inside it
        for _ in range(60):
            x0[:, 3] = np.where(np.abs(_alat_np(x0)) > 0.5 * a_max, 0.95 * x0[:, 3], x0[:, 3])
    x0[:, 5] = x0[:, 3] * np.interp(x0[:, 0], track.s_ref, track.kappa_ref)
    solver.set_x0(x0)
    solver.init_guess()
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 60.0 * np.arange(N)[None] / N; yref[:, :, 3] = 15.0
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 60.0; yref_e[:, 3] = 15.0
    solver.set_yref(yref); solver.set_yref_e(yref_e); solver.set_multipliers(None, None)
    return solver, P, x0, yref, yref_e


@pytest.mark.parametrize("soft,a_max", [(False, 3.0), (True, 2.0), (True, 5.0)])
def test_alat_row_rti_steps_match_oracle(track, soft, a_max):
    from oracle import oracle as orc

    B = 66
    s, P, x0, yref, yref_e = _setup(track, soft, B, 77, a_max)
    assert P.nc == 15
    x, u = s.get_x(), s.get_u()
    pi = lam = None
    n_active = 0
    for it in range(3):                     # three iterations on the frozen problem, multipliers carried along
        x_lin = x.copy()
        status = s.solve()
        out = P.rti_step(x, u, x0, yref, yref_e, pi=pi, lam=lam)
        pi, lam = out["pi"], out["lam"]
        np.testing.assert_array_equal(status, out["status"])
        ok = status == 0
        assert ok.sum() >= 0.9 * B
        np.testing.assert_array_equal(s.get_qp_iter()[ok], out["qp_iter"][ok])
        assert _rel(s.get_x()[ok], x[ok]) < 1e-6       # tolerance 1e-6 relative (north star: 1e-5)
        assert _rel(s.get_u()[ok], u[ok]) < 1e-6
        assert _rel(s.get_residuals(), out["res"]) < 1e-7
        _, lam_g = s.get_multipliers()
        lam_a, slk_a = s.get_alat_multipliers()
        lam30 = orc.widen_rows(lam_g, lam_a)
        assert np.max(np.abs(lam30[ok] - lam[ok])) / (1.0 + np.abs(lam).max()) < 1e-5
        assert np.all(lam_a[:, 0] == 0.0) and np.all(lam_a[:, N] == 0.0)        # stages 1..N-1 only
        n_active += int((np.abs(lam[ok][:, :, [14, 29]]) > 1e-3).sum())
        if soft:
            assert np.all(slk_a >= 0.0) and np.all(slk_a[:, 0] == 0.0) and np.all(slk_a[:, N] == 0.0)
        else:
            assert np.all(slk_a == 0.0)
            # the LINEARISED row is met by the step (a_lat itself only up to its curvature over the step)
            dz = s.get_x() - x_lin
            for b in np.flatnonzero(ok)[::7]:
                for k in range(1, N):
                    val, g = orc.alat(x_lin[b, k])
                    assert abs(val + g @ dz[b, k]) < a_max + 1e-6
        # keep both sides on the same iterate (differences of 1e-7 would otherwise compound through the non-convex NLP)
        s.set_x(x); s.set_u(u); s.set_multipliers(pi, np.ascontiguousarray(np.concatenate([lam[:, :, :14], lam[:, :, 15:29]], -1)))
        s.set_alat_multipliers(np.ascontiguousarray(lam[:, :, [14, 29]]), None)
    assert n_active > 0                         # the row really binds
    if soft and a_max < 3.0:
        assert slk_a.max() > 1e-3               # some car is beyond the tight bound and pays for it
    s.free()


@pytest.mark.parametrize("build", ["default", "ilp"])
@pytest.mark.parametrize("soft", [True, False])
def test_alat_instantiation_on_4096_instances_matches_oracle(track, soft, build):
    """`k_qp_wave<10,4,2,1>` on a large batch, in both scheduler builds: every status, every IPM iteration count of the solved instances,
    states, controls and residuals against the oracle."""
    from test_gpu_configs import _build

    B = 4096
    with _build(build):
        s, P, x0, yref, yref_e = _setup(track, soft, B, 4321, 2.5 if soft else 3.0)
    x, u = s.get_x(), s.get_u()
    status = s.solve()
    out = P.rti_step(x, u, x0, yref, yref_e)
    np.testing.assert_array_equal(status, out["status"])
    ok = status == 0
    assert ok.mean() > 0.9
    same = s.get_qp_iter()[ok] == out["qp_iter"][ok]
    assert same.mean() >= 0.999                 # (a marginal QP may stop an iteration apart: the canonical slot sums of round 4, NOTES.md)
    xs, us = s.get_x()[ok][same], s.get_u()[ok][same]
    assert _rel(xs, x[ok][same]) < 1e-6 and _rel(us, u[ok][same]) < 1e-6      # tolerance 1e-6 relative
    assert _rel(s.get_residuals(), out["res"]) < 1e-7
    lam_a, _ = s.get_alat_multipliers()
    assert (np.abs(lam_a[ok]) > 1e-3).any(axis=(1, 2)).mean() > 0.05        # the row binds on a fair share of the batch
    s.free()


def test_an_idle_row_gives_the_iterates_of_the_kernel_without_it(track):
    """Bounds of 1e4 m/s^2: `<10,4,2,1>` with a row that never binds against `<8,3,1,1>` without it -- same QP solution up to the QP tolerance."""
    B = 128
    res = []
    for row in (True, False):
        s, _, x0, _, _ = _setup(track, True, B, 5, 1e4, row=row, qp_tol=1e-10, qp_solver_iter_max=60)
        st = s.solve()
        res.append((st, s.get_x(), s.get_u()))
        if row:
            lam_a, slk_a = s.get_alat_multipliers()
            assert np.abs(lam_a).max() < 1e-5 and np.all(slk_a[st == 0] < 1e-6)
        s.free()
    ok = (res[0][0] == 0) & (res[1][0] == 0)
    assert ok.mean() > 0.9
    # (two barrier paths -- ten slots a lane against eight, and the idle row carries its mu / t^2 -- to the same solution)
    assert _rel(res[0][1][ok], res[1][1][ok]) < 1e-5 and _rel(res[0][2][ok], res[1][2][ok]) < 1e-5


def test_closed_loop_steps_with_the_row_launch_per_step_and_keep_it(track):
    """ihm2mpc_run_steps with the row: no persistent instantiation, so it launches per step -- same results as explicit step() calls --
    and the cars' lateral acceleration stays near the (soft) bound while the loop without the row goes beyond it."""
    B = 64
    a_max = 3.0
    worst = {}
    for row in (True, False):
        s, _, x0, _, _ = _setup(track, True, B, 9, a_max, row=row, penalty=1e3)
        s.solve(3)
        u_hist = s.run_steps(60.0, 12, model=0, M_sim=25, u0_hist=True)["u0"]
        if row:
            s2, _, _, _, _ = _setup(track, True, B, 9, a_max, row=True, penalty=1e3)
            s2.solve(3)
            for k in range(12):
                s2.step(60.0, model=0, M_sim=25)
                np.testing.assert_array_equal(s2.get_u0(), u_hist[k])
            s2.free()
        ok = s.get_status() == 0
        assert ok.mean() > 0.9
        worst[row] = np.maximum(np.abs(_alat_np(s.get_x()[ok])[:, 1:N]) - a_max, 0.0).sum(axis=1)      # a car's excess over the bound, summed over its horizon
        s.free()
    print("excess over a_max, median with / without the row:", np.median(worst[True]), np.median(worst[False]))
    assert np.median(worst[True]) < 0.5 * np.median(worst[False])


def test_refusals(track):
    from ihm2_amd.solver import BatchedOcpSolver

    def ocp_with_row(**kw):
        ocp = make_ocp(**kw)
        ocp.model.con_h_expr = "track+a_lat"
        c = ocp.constraints
        c.lh = np.array([-1e3, -1e3, -5.0]); c.uh = np.array([0.0, 0.0, 5.0]); c.lh_e = c.lh[:2]; c.uh_e = c.uh[:2]
        return ocp

    with pytest.raises(ValueError, match="kinematic"):
        ocp_with_row(model="fdyn6u").flatten()
    with pytest.raises(ValueError, match="SQP_RTI"):
        ocp_with_row(nlp_solver_type="SQP").flatten()
    # through the C ABI: the row without the track rows is refused at solve time with a message
    s = BatchedOcpSolver(make_ocp(), 8, track.s_ref, track.kappa_ref)
    from ihm2_amd import _lib
    assert s.lib.ihm2mpc_set_alat_constraint(s._h, 1, -5.0, 5.0, None, None) == 0
    s.set_x0(sample_x0(track, 8))
    with pytest.raises(RuntimeError, match="track rows"):
        s.init_guess()
    assert s.lib.ihm2mpc_set_alat_constraint(s._h, 1, 5.0, -5.0, None, None) != 0 and b"a_lat_min > a_lat_max" in s.lib.ihm2mpc_last_error()
    assert s.lib.ihm2mpc_set_alat_constraint(s._h, 0, 0.0, 0.0, None, None) == 0
    s.init_guess()
    assert np.all(s.solve() == s.get_status())
    s.free()


def test_controller_with_the_row_keeps_the_lateral_acceleration_down(track):
    """IHM2Controller(lateral_acceleration_row=True): ModelBounds.a_lat_max (python/mpc.py:26, unused by the live reference) becomes the bound
    of the row; a closed loop on the kinematic plant at a target speed the track's corners do not allow within 3 m/s^2."""
    from ihm2_amd.closed_loop_sim import SimModelVariant, Simulator, SimulatorConfig, run_closed_loop
    from ihm2_amd.controller import IHM2Controller

    B, steps, a_max = 32, 60, 3.0
    x0 = np.zeros((B, 8)); x0[:, 0] = np.linspace(0.0, 0.9 * track.lap_length, B); x0[:, 3] = 6.0
    w = np.array([[track.right_widths.min(), track.left_widths.min()]])
    peak = {}
    for row in (True, False):
        ctrl = IHM2Controller(track.s_ref, track.kappa_ref, batch_size=B, a_lat_max=a_max, track_widths=w, track_rows_penalty=(1e3, 1e3),
                              lateral_acceleration_row=row, s_target=50.0)
        assert ctrl.solver.data.alat_on == int(row)
        sim = Simulator(ctrl, SimulatorConfig(sampling_time=ctrl.dt, num_steps=40), SimModelVariant.KIN6)
        ctrl.warm_start(x0)
        res = run_closed_loop(ctrl, sim, x0, steps, lap_length=None)
        assert res.alive.mean() > 0.9
        xs = np.asarray(res.x)                      # (steps + 1, B, 8): the plant states the cars went through
        peak[row] = np.abs(_alat_np(xs[5:, res.alive])).max(axis=0)
        ctrl.solver.free()
    print("peak |a_lat| of the plant states, median with / without the row:", np.median(peak[True]), np.median(peak[False]))
    # (a soft row: the L1 + L2 penalty of 1e3 is weighed against the progress cost -- measured 4.8 against 45 m/s^2 on the kinematic plant)
    assert np.median(peak[True]) < a_max + 2.5 and np.median(peak[False]) > 3.0 * np.median(peak[True])
