"""GPU parity of the globalised SQP mode (python/main.py:230-237: "SQP", max_iter 2, "MERIT_BACKTRACKING") against the CPU
oracle's statement of the same algorithm (oracle/ihm2_oracle_rti.c, orc_sqp_solve)."""
import numpy as np
import pytest
from conftest import make_ocp, sample_x0

pytestmark = pytest.mark.gpu

N = 40


def _rel(a, b, floor=1.0):
    return np.max(np.abs(a - b) / (floor + np.abs(b)))


def _setup(track, model, variant, B, seed, **opts):
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import oracle as orc

    o = dict(nlp_solver_type="SQP", globalization="MERIT_BACKTRACKING", nlp_tol=1e-3, nlp_solver_tol_eq=1e-8, nlp_solver_tol_ineq=1e-8)
    o.update(opts)
    ocp = make_ocp(model=model, n_max=0.6 if variant == "soft" else 2.0, **o)
    c = ocp.constraints
    widths = None
    if variant == "soft":       # soft track bound on n (both sides) and soft steering-rate row
        c.idxsbx = np.array([0]); c.idxsg = np.array([1])
        ocp.cost.zl = np.array([50.0, 5.0]); ocp.cost.zu = np.array([50.0, 5.0])
        ocp.cost.Zl = np.array([200.0, 20.0]); ocp.cost.Zu = np.array([200.0, 20.0])
    if variant == "track":      # soft nonlinear track rows (old/generate_acaods_interface.py:380-449)
        ocp.model.con_h_expr = "track"
        c.lh = c.lh_e = np.array([-1e3, -1e3]); c.uh = c.uh_e = np.array([0.0, 0.0])
        c.idxsh, c.idxsh_e = np.arange(2), np.arange(2)
        ocp.cost.zl = ocp.cost.zu = ocp.cost.Zl = ocp.cost.Zu = np.full(2, 100.0)
        ocp.cost.zl_e = ocp.cost.zu_e = ocp.cost.Zl_e = ocp.cost.Zu_e = np.full(2, 100.0)
        widths = np.array([[1.2, 1.1]])
    data = ocp.flatten()
    solver = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref, track_widths=widths)
    if data.soft_Z is not None:
        solver.set_soft(data.soft_z, data.soft_Z)
    P = orc.OracleProblem(data.as_dict(track.s_ref, track.kappa_ref, track_widths=widths))
    x0 = sample_x0(track, B, seed=seed)
    if model != "fkin6":
        x0[:, 3] = np.linspace(6.0, 14.0, B)
    solver.set_x0(x0); solver.init_guess()
    u = solver.get_u()
    u[:, :, 1] = np.clip(u[:, :, 1] + 0.1 * np.sin(np.arange(N))[None], -0.5, 0.5)     # a poor steering guess: full steps overshoot
    solver.set_u(u)
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(N)[None] / N
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
    if variant == "track":
        yref[:, :, 1] = np.where(np.arange(B) % 2 == 0, 1.5, -1.5)[:, None]; yref_e[:, 1] = yref[:, 0, 1]
    solver.set_yref(yref); solver.set_yref_e(yref_e); solver.set_multipliers(None, None)
    return solver, P, data, x0, yref, yref_e


@pytest.mark.parametrize("model,variant,suff", [("fkin6", "hard", False), ("fkin6", "hard", True), ("fkin6", "soft", False),
                                                ("fdyn6u", "track", False)])
def test_sqp_iterations_with_line_search_match_oracle(track, model, variant, suff):
    """One SQP iteration per call, both sides re-synchronised in between (1e-7 differences would otherwise flip a merit
    comparison somewhere in the batch); the merit weights restart with every call on both sides."""
    B = 70
    s, P, data, x0, yref, yref_e = _setup(track, model, variant, B, 123, nlp_solver_max_iter=1, line_search_use_sufficient_descent=int(suff))
    x, u = s.get_x(), s.get_u()
    pi = np.zeros((B, N + 1, 8)); lam = np.zeros((B, N + 1, 28)); sl = np.zeros((B, N + 1, 28))
    n_short = 0
    for it in range(4):
        status = s.solve()
        out = P.sqp_solve(x, u, x0, yref, yref_e, pi=pi, lam=lam, sl=sl, max_iter=1, tol=data.sqp_tol, use_sufficient_descent=suff)
        st = s.get_sqp_stats()
        np.testing.assert_array_equal(status, out["status"])
        np.testing.assert_array_equal(st["sqp_iter"], out["sqp_iter"])
        np.testing.assert_array_equal(s.get_qp_iter(), out["qp_iter"])
        ok = np.isin(status, (0, 2))
        same = ok & (np.abs(st["alpha"] - out["alpha"]) < 1e-12)
        assert same.sum() >= 0.95 * ok.sum()           # a tie m(alpha) ~ m(0) at rounding level may fall either way
        n_short += int((out["alpha"][ok] < 1.0).sum())
        xg, ug = s.get_x(), s.get_u()
        assert _rel(xg[same], x[same]) < 1e-6          # tolerance 1e-6 relative (north star: 1e-5)
        assert _rel(ug[same], u[same]) < 1e-6
        np.testing.assert_allclose(s.get_u0()[same], ug[same][:, 0], rtol=0, atol=0)
        assert _rel(s.get_residuals(), out["res"]) < 1e-7
        pg, lg = s.get_multipliers()
        assert np.max(np.abs(lg[same] - lam[same])) / (1.0 + np.abs(lam).max()) < 1e-5
        assert np.max(np.abs(pg[same] - pi[same])) / (1.0 + np.abs(pi).max()) < 1e-5
        if data.soft_Z is not None:
            assert np.max(np.abs(s.get_slacks()[same] - sl[same])) < 1e-5 * (1.0 + np.abs(sl).max())
        # failed instances keep their iterate on both sides
        if (~ok).any():
            np.testing.assert_array_equal(xg[~ok], x[~ok])
        s.set_x(x); s.set_u(u); s.set_multipliers(pi, lam); s.set_slacks(sl)
        if model == "fkin6":
            assert ok.sum() >= 0.9 * B
    assert n_short >= 3                                 # the line search really shortened steps


def test_multi_iteration_sqp_solve_matches_oracle(track):
    """python/main.py:231: two SQP iterations inside one solve() -- merit weights carried from the first to the second."""
    B = 66
    s, P, data, x0, yref, yref_e = _setup(track, "fkin6", "hard", B, 7, nlp_solver_max_iter=2)
    x, u = s.get_x(), s.get_u()
    status = s.solve()
    out = P.sqp_solve(x, u, x0, yref, yref_e, max_iter=2, tol=data.sqp_tol)
    st = s.get_sqp_stats()
    assert np.mean(status == out["status"]) > 0.97 and np.all(np.isin(status, (0, 2, 4)))
    same = (status == out["status"]) & np.isin(status, (0, 2)) & (np.abs(st["alpha"] - out["alpha"]) < 1e-12) & (st["sqp_iter"] == out["sqp_iter"])
    assert same.sum() >= 0.85 * B
    # the first iteration's alpha is not reported: an early tie can only show up as a large difference, so compare the bulk
    dx = np.max(np.abs(s.get_x() - x) / (1.0 + np.abs(x)), axis=(1, 2))
    assert np.mean(dx[same] < 1e-6) > 0.95              # tolerance 1e-6 relative (north star: 1e-5)
    assert np.all(st["sqp_iter"][np.isin(status, (2,))] == 2)


def test_converged_instances_are_left_alone_and_report_status_0(track):
    B = 40
    # stationarity and complementarity are absolute, on gradients ~1e4: the QP tolerance (relative) bounds what can be reached
    s, P, data, x0, yref, yref_e = _setup(track, "fkin6", "hard", B, 99, nlp_solver_max_iter=20, qp_tol=1e-8, qp_solver_iter_max=60,
                                          nlp_tol=1e-2, nlp_solver_tol_eq=1e-7, nlp_solver_tol_ineq=1e-7)
    status = s.solve()
    st = s.get_sqp_stats()
    conv = status == 0
    assert conv.sum() >= B // 4 and np.all(st["sqp_iter"][conv] < 20) and np.all(st["sqp_iter"][status == 2] == 20)
    res = s.get_residuals()
    assert np.all(res[conv] <= data.sqp_tol[None] * (1 + 1e-12))
    x1, u1 = s.get_x(), s.get_u()
    status2 = s.solve()
    st2 = s.get_sqp_stats()
    assert np.all(status2[conv] == 0) and np.all(st2["sqp_iter"][conv] == 0)
    np.testing.assert_array_equal(s.get_x()[conv], x1[conv]); np.testing.assert_array_equal(s.get_u()[conv], u1[conv])
    # per-instance shim: acados' get_stats names
    from ihm2_amd.solver import AcadosOcpSolver
    i = int(np.flatnonzero(conv)[0])
    view = s[i] if hasattr(s, "__getitem__") else AcadosOcpSolver(s, i)
    assert view.get_stats("sqp_iter") == 0 and view.get_status() == 0


@pytest.mark.parametrize("model", ["fkin6", "fdyn6u"])
def test_live_solver_options_sqp_with_irk_match_oracle(track, model):
    """python/main.py:227-238 as the reference runs it: SQP, two iterations, MERIT_BACKTRACKING, IRK with four Gauss-Legendre stages and
    one step per interval.  The line search's trial points are integrated by k_rollout_irk (the first three step lengths for all
    instances, the rest of the ladder for the instances the first line-search launch leaves open); one SQP iteration per call with both
    sides re-synchronised in between, as above."""
    B = 66
    s, P, data, x0, yref, yref_e = _setup(track, model, "hard", B, 31, nlp_solver_max_iter=1, integrator_type="IRK", sim_method_num_steps=1)
    assert data.integrator != 0
    x, u = s.get_x(), s.get_u()
    pi = np.zeros((B, N + 1, 8)); lam = np.zeros((B, N + 1, 28)); sl = np.zeros((B, N + 1, 28))
    n_short = n_deep = 0
    for it in range(3):
        status = s.solve()
        out = P.sqp_solve(x, u, x0, yref, yref_e, pi=pi, lam=lam, sl=sl, max_iter=1, tol=data.sqp_tol)
        st = s.get_sqp_stats()
        assert np.mean(status == out["status"]) >= 0.97
        both = (status == out["status"]) & np.isin(status, (0, 2))
        np.testing.assert_array_equal(s.get_qp_iter()[both], out["qp_iter"][both])
        same = both & (np.abs(st["alpha"] - out["alpha"]) < 1e-12)
        assert same.sum() >= 0.95 * both.sum()
        n_short += int((out["alpha"][both] < 1.0).sum())
        n_deep += int((out["alpha"][same] < 0.7 ** 2 - 1e-9).sum())      # past the third trial: settled by the second pair of launches
        assert _rel(s.get_x()[same], x[same]) < 1e-6 and _rel(s.get_u()[same], u[same]) < 1e-6          # tolerance 1e-6 relative
        s.set_x(x); s.set_u(u); s.set_multipliers(pi, lam); s.set_slacks(sl)
        if model == "fkin6":
            assert both.sum() >= 0.9 * B
    assert n_short >= 1
    assert n_deep >= 1          # the second pair of launches is exercised (and agreed with the oracle's step length to 1e-12)
    # two iterations inside one solve (the live max_iter = 2): runs, and every instance ends in an accepted status or a reported failure
    s2, _, _, _, _, _ = _setup(track, model, "hard", B, 31, nlp_solver_max_iter=2, integrator_type="IRK", sim_method_num_steps=1)
    st2 = s2.solve()
    assert np.all(np.isin(st2, (0, 1, 2, 4))) and np.isin(st2, (0, 2)).mean() >= (0.9 if model == "fkin6" else 0.5)


def test_long_backtracking_ladder_with_irk_matches_oracle_and_over_long_ladders_are_refused(track):
    """alpha_reduction close to 1 makes a long ladder of trial steps.  The collocation path keeps one rollout per step length of
    the ladder in a buffer: 59 step lengths (rho = 0.95, alpha_min = 0.05) run and agree with the oracle's step lengths; a ladder beyond the
    buffer's limit of 64 (rho = 0.97: 99 step lengths) is refused by ihm2mpc_set_sqp_options instead of running past the buffer."""
    B = 40
    s, P, data, x0, yref, yref_e = _setup(track, "fkin6", "hard", B, 7, nlp_solver_max_iter=1, integrator_type="IRK", sim_method_num_steps=1)
    s.set_sqp_options(alpha_min=0.05, alpha_reduction=0.95, tol=data.sqp_tol)
    x, u = s.get_x(), s.get_u()
    pi = np.zeros((B, N + 1, 8)); lam = np.zeros((B, N + 1, 28)); sl = np.zeros((B, N + 1, 28))
    n_short = 0
    for it in range(2):
        status = s.solve()
        out = P.sqp_solve(x, u, x0, yref, yref_e, pi=pi, lam=lam, sl=sl, max_iter=1, tol=data.sqp_tol, alpha_min=0.05, alpha_reduction=0.95)
        st = s.get_sqp_stats()
        both = (status == out["status"]) & np.isin(status, (0, 2))
        assert both.sum() >= 0.9 * B
        same = both & (np.abs(st["alpha"] - out["alpha"]) < 1e-12)
        assert same.sum() >= 0.9 * both.sum()          # a merit tie at rounding level may fall one rung either way
        assert _rel(s.get_x()[same], x[same]) < 1e-6 and _rel(s.get_u()[same], u[same]) < 1e-6          # tolerance 1e-6 relative
        n_short += int((out["alpha"][same] < 1.0).sum())
        s.set_x(x); s.set_u(u); s.set_multipliers(pi, lam); s.set_slacks(sl)
    assert n_short >= 1
    with pytest.raises(RuntimeError, match="ladder"):
        s.set_sqp_options(alpha_min=0.05, alpha_reduction=0.97)
    s.set_sqp_options(globalization="FIXED_STEP", alpha_min=0.05, alpha_reduction=0.97)        # no ladder without the line search
