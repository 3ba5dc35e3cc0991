"""GPU parity / property tests of the BASELINE.json configurations that round 1 left untested:

* the benchmarked entry point itself -- the persistent per-instance loop ``ihm2mpc_run_steps`` -- directly against an oracle loop
  (not only bit-equality with the per-step path);
* configs[3]'s workload: dynamic bicycle (as written and un-crossed), per-instance ``track_id`` over every track of ``data/``
  the table builder takes, against the oracle;
* full-size property runs of configs[1] (B = 1024 kinematic) and configs[2] (B = 8192 dynamic + soft track rows): statuses, NLP KKT
  residuals of the QP that was solved, bounds respected by the iterate.

Floating point: tolerances are written at each assert (north star: 1e-5 relative on state / control)."""
import numpy as np
import pytest
from conftest import make_ocp, sample_x0

pytestmark = pytest.mark.gpu

N = 40
ALL_TRACKS = ("fsds_competition_1", "fsds_competition_2", "fsds_competition_3", "fsds_default", "acceleration", "skidpad", "short_skidpad")


def _rel(a, b, floor=1.0):
    return float(np.max(np.abs(a - b) / (floor + np.abs(b)))) if a.size else 0.0


def test_persistent_loop_matches_oracle_loop(track):
    """ihm2mpc_run_steps (the kernel bench.py times): plant -> shift + ramp -> linearise -> QP, 12 steps of 96 instances in ONE
    launch, against the same loop driven by the oracle: controls, plant states, statuses and IPM iteration counts of every step."""
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import oracle as orc

    B, steps, M_sim = 96, 12, 25
    ocp = make_ocp()
    s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, B, seed=2024)
    s.set_x0(x0); s.init_guess()
    x, u = s.get_x(), s.get_u()
    # a first solve without the plant on both sides: the loop starts from a control
    s.prepare_step(40.0); st = s.solve()
    yref, yref_e = orc.prepare_step(N, x0, 40.0, x, u)
    out = P.rti_step(x, u, x0, yref, yref_e)
    np.testing.assert_array_equal(st, out["status"])
    pi, lam = out["pi"], out["lam"]
    h = s.run_steps(40.0, steps, model=0, M_sim=M_sim, u0_hist=True, x0_hist=True, status_hist=True, qp_iter_hist=True)
    xc = x0.copy()
    for i in range(steps):
        xc = P.sim_step(xc, u[:, 0].copy(), 0, M_sim)
        yref, yref_e = orc.prepare_step(N, xc, 40.0, x, u)
        out = P.rti_step(x, u, xc, yref, yref_e, pi=pi, lam=lam)
        pi, lam = out["pi"], out["lam"]
        assert _rel(h["x0"][i], xc) < 1e-7, i                   # plant state the step was solved from (tolerance 1e-7 relative)
        np.testing.assert_array_equal(h["status"][i], out["status"])
        assert np.all(out["status"] == 0)
        assert np.mean(h["qp_iter"][i] == out["qp_iter"]) >= 0.97, i      # a marginal convergence test may fall either way after several steps
        assert _rel(h["u0"][i], u[:, 0]) < 1e-6, i              # tolerance 1e-6 relative (north star: 1e-5)
    assert _rel(s.get_x(), x) < 1e-6 and _rel(s.get_u(), u) < 1e-6


def test_long_closed_loop_stays_with_the_oracle(track):
    """Drift check over the length of a bench run and beyond: 150 control steps (7.5 s of driving, lap wrap included) of 48 instances in
    persistent launches of 25 steps, against the oracle loop.  Measured (tools/dbg_drift.py): controls and plant states agree to 1e-9 all
    along, except where a marginal convergence test falls the other way on one side -- visibly (an iteration count that differs) or not (the
    followed residuals pass and their check against the data does or does not, so the last Newton step takes the residuals from another
    source): the two controls are then one interior-point tolerance apart (3e-5 observed; 3e-4 rad is the solver's accuracy at tol 1e-6 of
    the gradient scale) for ONE step and the stable closed loop is back at 1e-9 two steps later.  Asserted: at least
    99 % of the (instance, step) pairs within 1e-6 relative (north star: 1e-5), every pair within 3e-4, statuses equal, no drift at the end."""
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import oracle as orc

    B, M_sim, L = 48, 25, track.lap_length
    ocp = make_ocp()
    s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, B, seed=77)
    s.set_x0(x0); s.init_guess(); s.set_lap_wrap(True)
    x, u = s.get_x(), s.get_u()
    s.prepare_step(40.0); st = s.solve()
    yref, yref_e = orc.prepare_step(N, x0, 40.0, x, u)
    out = P.rti_step(x, u, x0, yref, yref_e)
    np.testing.assert_array_equal(st, out["status"])
    pi, lam = out["pi"], out["lam"]
    xc = x0.copy()
    same_iters = total = close = 0
    good = np.ones(B, dtype=bool)          # instances that have solved every step so far
    dev = lambda a, b: np.max(np.abs(a - b) / (1.0 + np.abs(b)), axis=-1)
    for chunk in range(6):
        h = s.run_steps(40.0, 25, model=0, M_sim=M_sim, u0_hist=True, x0_hist=True, status_hist=True, qp_iter_hist=True)
        for i in range(25):
            w = xc[:, 0] >= L                                # ihm2mpc_set_lap_wrap: a car past the lap is moved back by one lap, iterate included
            xc[w, 0] -= L; x[w, :, 0] -= L
            xc = P.sim_step(xc, u[:, 0].copy(), 0, M_sim)
            yref, yref_e = orc.prepare_step(N, xc, 40.0, x, u)
            out = P.rti_step(x, u, xc, yref, yref_e, pi=pi, lam=lam)
            pi, lam = out["pi"], out["lam"]
            # a failed instance keeps its iterate on both sides and is left out from then on
            np.testing.assert_array_equal(h["status"][i], out["status"])
            good &= out["status"] == 0
            d = np.maximum(dev(h["u0"][i], u[:, 0]), dev(h["x0"][i], xc))[good]
            assert np.all(d < 3e-4), (chunk, i, float(d.max()))
            close += int(np.sum(d < 1e-6)); total += int(good.sum())
            same_iters += int(np.sum((h["qp_iter"][i] == out["qp_iter"])[good]))
    assert close >= 0.99 * total and same_iters >= 0.97 * total and good.sum() >= B - 2
    assert _rel(s.get_x()[good], x[good]) < 1e-6 and _rel(s.get_u()[good], u[good]) < 1e-6          # no drift left at the end
    assert np.any(xc[:, 0] < x0[:, 0])          # somebody completed the lap and was wrapped


def _accepted_tracks():
    """Every data/ track the table builder takes (three of the seven are open paths)."""
    from ihm2_amd.track import track_table

    ok, why = [], {}
    for name in ALL_TRACKS:
        try:
            ok.append((name, track_table(name)))
        except Exception as e:      # noqa: BLE001 -- the reason is reported by the test
            why[name] = repr(e)
    return ok, why


@pytest.mark.parametrize("model", ["fdyn6", "fdyn6u"])
def test_config3_dynamic_model_over_all_tracks_matches_oracle(model):
    """BASELINE.json configs[3] (the per-GPU share, small): dynamic bicycle, instance b on track b mod T over every track of data/
    the table builder accepts, soft nonlinear track rows, against the oracle with the same per-instance track ids."""
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import oracle as orc

    plans, why = _accepted_tracks()
    assert len(plans) >= 4, why                       # at least the four closed FSDS tracks
    names = [n for n, _ in plans]
    print("tracks accepted:", names, "rejected:", why)
    B = 12 * len(plans)
    ocp = make_ocp(model=model)
    ocp.model.con_h_expr = "track"
    c = ocp.constraints
    c.lh = c.lh_e = np.array([-1e3, -1e3]); c.uh = c.uh_e = np.array([0.0, 0.0])
    c.idxsh, c.idxsh_e = np.arange(2), np.arange(2)
    ocp.cost.zl = ocp.cost.zu = ocp.cost.Zl = ocp.cost.Zu = np.full(2, 100.0)
    ocp.cost.zl_e = ocp.cost.zu_e = ocp.cost.Zl_e = ocp.cost.Zu_e = np.full(2, 100.0)
    s_ref = np.stack([p.s_ref for _, p in plans]); k_ref = np.stack([p.kappa_ref for _, p in plans])
    widths = np.array([[p.right_widths.min(), p.left_widths.min()] for _, p in plans])
    tid = (np.arange(B) % len(plans)).astype(np.int32)
    s = BatchedOcpSolver(ocp, B, s_ref, k_ref, track_id=tid, track_widths=widths)
    P = orc.OracleProblem(ocp.flatten().as_dict(s_ref, k_ref, track_widths=widths))
    x0 = np.zeros((B, 8))
    for t, (_, p) in enumerate(plans):
        sel = tid == t
        x0[sel] = sample_x0(p, int(sel.sum()), seed=300 + t)
    x0[:, 3] = np.clip(x0[:, 3], 4.0, 12.0)
    s.set_x0(x0); s.init_guess()
    x, u = s.get_x(), s.get_u()
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(N)[None] / N
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
    s.set_yref(yref); s.set_yref_e(yref_e); s.set_multipliers(None, None)
    pi = lam = None
    for it in range(2):
        st = s.solve()
        out = P.rti_step(x, u, x0, yref, yref_e, track_id=tid, pi=pi, lam=lam)
        pi, lam = out["pi"], out["lam"]
        # one code per outcome on both sides (a diverging QP is a failed QP, 4); the open-loop unstable model as written amplifies rounding
        # differences until a few marginal QPs converge on one side only
        same = st == out["status"]
        assert same.mean() >= 0.95, (it, st, out["status"])
        ok = (st == 0) & (out["status"] == 0)
        assert ok.sum() >= (0.3 * B if model == "fdyn6" else 0.8 * B), (it, np.bincount(st, minlength=5))
        for t in range(len(plans)):      # every track contributes solved instances
            assert (ok & (tid == t)).sum() >= 1, (names[t], st[tid == t])
        assert _rel(s.get_x()[ok], x[ok]) < 1e-6 and _rel(s.get_u()[ok], u[ok]) < 1e-6      # tolerance 1e-6 relative
        if it == 0:      # identical inputs on both sides (afterwards the iterates agree to 1e-6 only and the residuals cancel digits)
            # (an initial rollout that leaves the track by several metres gets near the singularity 1 + kappa n = 0 of the Frenet
            # frame, where rounding differences of the integrators are amplified 1e10-fold: such instances are not compared)
            sane = same & (np.abs(x[:, :, 1]).max(axis=1) < 3.0)
            assert sane.mean() > 0.9
            rg, ro = s.get_residuals()[sane], out["res"][sane]
            dev = np.max(np.abs(rg - ro) / (1.0 + np.abs(ro)), axis=0)
            assert np.all(dev < 1e-7), ("relative deviation of (stat, eq, ineq, comp)", dev)
        # keep the two sides together where only one of them failed
        bad = ~ok
        if bad.any():
            xg, ug = s.get_x(), s.get_u()
            x[bad], u[bad] = xg[bad], ug[bad]
            pg, lg = s.get_multipliers()
            pi[bad], lam[bad] = pg[bad], lg[bad]
            s.set_x(x); s.set_u(u); s.set_multipliers(pi, lam)


def _check_iterate(solver, ocp, B, frac_ok):
    st = solver.get_status()
    ok = st == 0
    assert ok.mean() >= frac_ok, np.bincount(st, minlength=5)
    x, u = solver.get_x(), solver.get_u()
    assert np.all(np.isfinite(x[ok])) and np.all(np.isfinite(u[ok]))
    c = ocp.constraints
    tol = 1e-6
    # hard boxes hold at the new iterate (the QP step is feasible for the linear rows; x_0 is the measured state)
    for i, lo, hi in zip(c.idxbx, c.lbx, c.ubx):
        assert np.all(x[ok][:, 1:-1, i] >= lo - tol * (1 + abs(lo))) and np.all(x[ok][:, 1:-1, i] <= hi + tol * (1 + abs(hi))), i
    for i, lo, hi in zip(c.idxbu, c.lbu, c.ubu):
        assert np.all(u[ok][:, :, i] >= lo - tol * (1 + abs(lo))) and np.all(u[ok][:, :, i] <= hi + tol * (1 + abs(hi))), i
    g = u[ok] - x[ok][:, :-1, 6:8]                       # rate rows C x + D u (python/mpc.py:92-99)
    assert np.all(g >= c.lg - tol * (1 + np.abs(c.lg))) and np.all(g <= c.ug + tol * (1 + np.abs(c.ug)))
    res = solver.get_residuals()[ok]                      # NLP KKT inf-norms at the iterate the QP was built at
    assert np.all(np.isfinite(res))
    return ok


def test_config1_full_size_properties(track):
    """configs[1] at full size (B = 1024 kinematic, N = 40): 10 control steps in the persistent loop; every instance solves,
    bounds hold, residuals are finite."""
    from ihm2_amd.solver import BatchedOcpSolver

    B = 1024
    ocp = make_ocp()
    s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
    x0 = sample_x0(track, B)
    s.set_x0(x0); s.init_guess(); s.set_lap_wrap(True)
    s.step(40.0, model=0, M_sim=25)
    h = s.run_steps(40.0, 10, model=0, M_sim=25, u0_hist=True, status_hist=True, qp_iter_hist=True)
    assert np.all(h["status"] == 0)
    assert np.all(np.abs(h["u0"][..., 0]) <= 500.0 * (1 + 1e-9)) and np.all(np.abs(h["u0"][..., 1]) <= 0.5 * (1 + 1e-9))
    assert 5 <= h["qp_iter"].mean() <= 15
    _check_iterate(s, ocp, B, 1.0)
    assert np.all(np.isfinite(s.get_residuals()))
    # the QPs' own KKT residuals at the returned points: every status-0 solve is inside the interior-point tolerance (relative 1e-6)
    q = s.get_qp_residuals()
    assert q.shape == (B, 4) and np.all(q >= 0.0) and np.all(q <= ocp.solver_options.qp_tol)


@pytest.mark.parametrize("model,frac", [("fdyn6u", 0.85), ("fdyn6", 0.03)])
def test_config2_full_size_properties(track, model, frac):
    """configs[2] at full size (B = 8192 dynamic bicycle + soft nonlinear track rows): 3 control steps with launches per step.
    The model as written (quirk Q3) is open-loop unstable: most of its QPs fail (reported), the solved ones
    must still satisfy the properties; the un-crossed model solves the bulk."""
    from ihm2_amd.solver import BatchedOcpSolver

    B = 8192
    ocp = make_ocp(model=model)
    ocp.model.con_h_expr = "track"
    c = ocp.constraints
    c.lh = c.lh_e = np.array([-1e3, -1e3]); c.uh = c.uh_e = np.array([0.0, 0.0])
    c.idxsh, c.idxsh_e = np.arange(2), np.arange(2)
    ocp.cost.zl = ocp.cost.zu = ocp.cost.Zl = ocp.cost.Zu = np.full(2, 100.0)
    ocp.cost.zl_e = ocp.cost.zu_e = ocp.cost.Zl_e = ocp.cost.Zu_e = np.full(2, 100.0)
    widths = np.array([[track.right_widths.min(), track.left_widths.min()]])
    s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref, track_widths=widths)
    x0 = sample_x0(track, B)
    s.set_x0(x0); s.init_guess()
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(N)[None] / N
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
    s.set_yref(yref); s.set_yref_e(yref_e); s.solve(3)
    plant = {"fdyn6": 1, "fdyn6u": 2}[model]
    for _ in range(3):
        s.step(40.0, model=plant, M_sim=25)
        s.synchronize()
    ok = _check_iterate(s, ocp, B, frac)
    slk = s.get_slacks()
    assert np.all(slk[ok] >= 0.0) and np.all(np.isfinite(slk[ok]))


# ---- no result may depend on the instruction order one compiler scheduler happens to pick ----
# `make -C ihm2_amd/csrc ilp` (run by __graft_entry__.build()) builds the same sources with LLVM's iterative ILP scheduler on the QP
# objects into ihm2_amd/libihm2mpc_ilp.so; round 1's `<8,3,1,1>` returned wrong statuses in such a build.
import contextlib
import os

from ihm2_amd import _lib as _libmod

ILP_LIB = os.path.join(os.path.dirname(_libmod.LIB_PATH), "libihm2mpc_ilp.so")


@contextlib.contextmanager
def _build(which):
    """Solvers created inside use the named build of the library (they keep their own reference to it afterwards)."""
    if which == "default":
        yield
        return
    if not os.path.exists(ILP_LIB):
        pytest.skip("libihm2mpc_ilp.so not built (make -C ihm2_amd/csrc ilp)")
    saved = (_libmod._lib, _libmod.LIB_PATH)
    _libmod._lib, _libmod.LIB_PATH = None, ILP_LIB
    try:
        yield
    finally:
        _libmod._lib, _libmod.LIB_PATH = saved


@pytest.mark.parametrize("build", ["default", "ilp"])
@pytest.mark.parametrize("soft", [True, False])
def test_track_row_instantiations_on_4096_instances_match_oracle(track, soft, build):
    """The track-row instantiations of the QP kernel (`<8,3,1,1>` soft, `<8,2,1,1>` hard) on a large batch, in both builds: every
    status, every IPM iteration count of the solved instances, states and controls against the oracle."""
    from test_gpu_parity import _path_setup

    B = 4096
    with _build(build):
        s, P, x0, yref, yref_e = _path_setup(track, "fkin6", soft, B, 1234, 1.2 if soft else 1.6)
    x, u = s.get_x(), s.get_u()
    status = s.solve()
    out = P.rti_step(x, u, x0, yref, yref_e)
    np.testing.assert_array_equal(status, out["status"])
    ok = status == 0
    assert ok.mean() > 0.9
    np.testing.assert_array_equal(s.get_qp_iter()[ok], out["qp_iter"][ok])
    assert _rel(s.get_x()[ok], x[ok]) < 1e-6 and _rel(s.get_u()[ok], u[ok]) < 1e-6      # tolerance 1e-6 relative
    assert _rel(s.get_residuals(), out["res"]) < 1e-7


def _soft_ocp(kind):
    """OCPs that select the instantiations without track rows: all-hard `<5,0,0>`; soft sides on n, v_x and the steering-rate row
    (the OCP of test_gpu_parity.py::setup_soft) for the soft tables `<8,2,0>` / `<10,4,0>`."""
    if kind == "hard":
        return make_ocp()
    ocp = make_ocp(n_max=0.3)           # the sampled |n| <= 0.5 violates the track bound: slacks are active
    c = ocp.constraints
    c.idxsbx = np.array([0, 1]); c.idxsg = np.array([1]); c.idxsbx_e = np.array([0])
    ocp.cost.zl = ocp.cost.zu = np.array([50.0, 10.0, 5.0]); ocp.cost.Zl = ocp.cost.Zu = np.array([200.0, 0.0, 20.0])
    ocp.cost.zl_e = ocp.cost.zu_e = np.array([50.0]); ocp.cost.Zl_e = ocp.cost.Zu_e = np.array([200.0])
    return ocp


@pytest.mark.parametrize("kind", ["hard", "soft_rows"])
def test_both_builds_agree_on_4096_instances(track, kind):
    """Per-step kernels and the persistent loop (`k_qp_wave`, `k_steps`) of the two builds: same statuses, same iteration counts,
    controls equal to 1e-9 (the arithmetic is the same, only its order in time may differ), and the default build against the
    oracle."""
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import oracle as orc

    B = 4096
    res = {}
    for build in ("default", "ilp"):
        ocp = _soft_ocp(kind)
        with _build(build):
            s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
        x0 = sample_x0(track, B, seed=77)
        s.set_x0(x0); s.init_guess(); s.set_multipliers(None, None)
        yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(N)[None] / N
        yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
        s.set_yref(yref); s.set_yref_e(yref_e)
        x, u = s.get_x(), s.get_u()
        st = s.solve()
        r = dict(status=st.copy(), it=s.get_qp_iter().copy(), u=s.get_u().copy())
        if build == "default":
            P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
            out = P.rti_step(x, u, x0, yref, yref_e)
            np.testing.assert_array_equal(st, out["status"])
            ok = st == 0
            assert ok.mean() > 0.95
            # (two summation orders of the barrier parameter: one marginal convergence test in a few thousand may fall the other way)
            same = r["it"] == out["qp_iter"]
            assert np.mean(same[ok]) >= 0.999 and np.max(np.abs(r["it"][ok] - out["qp_iter"][ok])) <= 1
            assert _rel(r["u"][ok & same], u[ok & same]) < 1e-6                           # tolerance 1e-6 relative (where both sides stopped at the same iterate)
        s.set_lap_wrap(True)
        h = s.run_steps(40.0, 4, model=0, M_sim=25, u0_hist=True, status_hist=True, qp_iter_hist=True)
        r.update(h_u0=h["u0"].copy(), h_st=h["status"].copy(), h_it=h["qp_iter"].copy())
        res[build] = r
    a, b = res["default"], res["ilp"]
    np.testing.assert_array_equal(a["status"], b["status"]); np.testing.assert_array_equal(a["it"], b["it"])
    np.testing.assert_array_equal(a["h_st"], b["h_st"]); np.testing.assert_array_equal(a["h_it"], b["h_it"])
    okm = a["status"] == 0
    assert _rel(a["u"][okm], b["u"][okm]) < 1e-9
    okh = np.all(a["h_st"] == 0, axis=0)                      # histories are (n_steps, B, .)
    assert okh.mean() > 0.95
    assert _rel(a["h_u0"][:, okh], b["h_u0"][:, okh]) < 1e-9


# ---- the per-GPU shares of configs[3] and configs[4] at full size, through the workloads bench.py runs ----
def test_config3_per_gpu_share_on_all_seven_tracks():
    """configs[3]'s share of one of 8 GPUs -- 8192 dynamic bicycles (un-crossed model), instance g on track g mod 7 over ALL seven tracks
    of data/, soft track rows, failed instances re-initialised -- five closed-loop steps after three of warm-up: at least 90 % of the
    solves succeed, every track contributes solved instances, and the gathered (u0, status) is what the handle holds."""
    import bench
    from ihm2_amd.dist import RankContext

    ranks = RankContext()               # one rank: no exchange, same code path as the sharded run
    ranks.total = 8192
    r = bench.rti_throughput(model="fdyn6u", B=8192, steps=5, warmup=3, tracks=bench.ALL_TRACKS, terminal_bounds="stage", track_rows="soft",
                             recover=True, ranks=ranks)
    assert r["tracks"] == 7 and r["ok_fraction"] >= 0.90, r
    u0, st = r["gathered"]
    assert u0.shape == (8192, 2) and st.shape == (8192,)
    tid = np.arange(8192) % 7
    for t in range(7):
        assert np.mean(st[tid == t] == 0) >= 0.80, (bench.ALL_TRACKS[t], np.bincount(st[tid == t], minlength=5))
    okm = st == 0
    assert np.all(np.abs(u0[okm, 0]) <= 500.0 * (1 + 1e-9)) and np.all(np.abs(u0[okm, 1]) <= 0.5 * (1 + 1e-9))


def test_config4_full_size_closed_loop_and_per_gpu_share():
    """configs[4]: 4096 cars x 200 control periods device-resident (one ihm2mpc_step per period), and the share of one of 8 GPUs -- 512 cars --
    both device-resident and in ONE persistent launch: the cars keep driving (alive + finished), and the two runners of the share agree car
    by car (same freezing rules on the host and on the device)."""
    import bench

    kw = dict(steps=200, terminal_bounds="stage", plant="KIN6_DYN6U", soft_state_bounds=(1000.0, 1000.0))
    full = bench.closed_loop_config5(B=4096, device_loop=True, **kw)
    assert full["steps"] == 200 and full["alive"] + full["finished"] >= 4080, full
    dev = bench.closed_loop_config5(B=512, device_loop=True, **kw)
    per = bench.closed_loop_config5(B=512, persistent=True, **kw)
    assert dev["alive"] + dev["finished"] == 512 and per["alive"] + per["finished"] == 512, (dev, per)
    assert (dev["alive"], dev["finished"], dev["failed"]) == (per["alive"], per["finished"], per["failed"])


def test_config4_full_size_with_the_plant_integrator_of_the_reference():
    """configs[4] with the plants as the reference configures them (python/main.py:395-400: IRK, GAUSS_RADAU_IIA, 4 stages, 100 steps per control
    period; `bench.py --config 4 --plant-integrator IRK`): the full 4096 cars device-resident and the share of one GPU in one persistent launch
    (Radau IIA x 100 inside k_steps) -- the same outcome as with RK4 x 100 up to a handful of cars, both runners of the share car by car."""
    import bench

    kw = dict(steps=200, terminal_bounds="stage", plant="KIN6_DYN6U", soft_state_bounds=(1000.0, 1000.0), plant_integrator="IRK")
    full = bench.closed_loop_config5(B=4096, device_loop=True, **kw)
    assert full["steps"] == 200 and full["alive"] + full["finished"] >= 4080, full
    dev = bench.closed_loop_config5(B=512, device_loop=True, **kw)
    per = bench.closed_loop_config5(B=512, persistent=True, **kw)
    assert dev["alive"] + dev["finished"] >= 508 and per["alive"] + per["finished"] >= 508, (dev, per)
    assert (dev["alive"], dev["finished"], dev["failed"]) == (per["alive"], per["finished"], per["failed"])
    assert abs(dev["progress_m_median"] - per["progress_m_median"]) <= 1e-6 * (1 + abs(dev["progress_m_median"]))
