"""SURVEY.md section 8f N1 on the device: the track tables (python/motion_planning.py:139-289, 345-428) built by
``ihm2mpc_build_tracks`` from the centre lines' spline coefficients, against the independent restatement ``oracle/track_np.py``
(null-space spline fit, point-by-point sampling: tests/test_oracle_track.py pins it) -- and, as a second check, the product's host planner."""
import numpy as np
import pytest

from conftest import make_ocp, sample_x0

pytestmark = pytest.mark.gpu


def test_device_track_tables_match_the_host_planner_on_all_tracks():
    from ihm2_amd import track as T
    from ihm2_amd.solver import BatchedOcpSolver

    names = T.TRACK_NAMES
    plans = [T.track_table(n) for n in names]
    cs = [T.fit_spline(T.load_track_geometry_data(n).center_line, curv_weight=2.0) for n in names]
    nk = plans[0].s_ref.size
    assert nk == 3 * T.NUMBER_SPLINE_INTERVALS
    dummy = np.tile(np.linspace(0.0, 1.0, nk), (len(names), 1))
    s = BatchedOcpSolver(make_ocp(), 70, dummy, np.zeros_like(dummy), track_id=np.arange(70) % len(names))
    s.build_tracks([c[0] for c in cs], [c[1] for c in cs])
    s_ref, kappa, X, Y, phi = s.get_tracks()
    from oracle import track_np as OT
    for t, n in enumerate(names):           # the checker is the oracle's own pipeline, fit included (not product code)
        geo = T.load_track_geometry_data(n)
        mp = OT.motion_plan(geo.center_line, geo.track_widths)
        L = mp["lap_length"]
        assert np.max(np.abs(s_ref[t] - mp["s_ref"])) < 1e-7 * L                                # tolerance 1e-7 of a lap (two different fits)
        assert np.max(np.abs(X[t] - mp["X_ref"])) < 1e-7 * L and np.max(np.abs(Y[t] - mp["Y_ref"])) < 1e-7 * L
        assert np.max(np.abs(kappa[t] - mp["kappa_ref"])) < 1e-6 * max(1.0, np.max(np.abs(mp["kappa_ref"])))
        assert np.max(np.abs(np.angle(np.exp(1j * (phi[t] - mp["phi_ref"]))))) < 1e-6
        # the device kernels alone (same coefficients in): 1e-9
        mp2 = OT.motion_plan(geo.center_line, geo.track_widths, coeffs=cs[t])
        assert np.max(np.abs(s_ref[t] - mp2["s_ref"])) < 1e-9 * L and np.max(np.abs(X[t] - mp2["X_ref"])) < 1e-9 * L
        assert np.max(np.abs(kappa[t] - mp2["kappa_ref"])) < 1e-9 * max(1.0, np.max(np.abs(mp2["kappa_ref"])))
    for t, p in enumerate(plans):
        L = p.lap_length
        assert np.max(np.abs(s_ref[t] - p.s_ref)) < 1e-9 * L                                    # tolerance 1e-9 of a lap
        assert np.max(np.abs(X[t] - p.X_ref)) < 1e-9 * L and np.max(np.abs(Y[t] - p.Y_ref)) < 1e-9 * L
        assert np.max(np.abs(kappa[t] - p.kappa_ref)) < 1e-9 * max(1.0, np.max(np.abs(p.kappa_ref)))
        assert np.max(np.abs(np.angle(np.exp(1j * (phi[t] - p.phi_ref))))) < 1e-9
        assert np.all(np.diff(s_ref[t]) > 0)
    # the tables are live: a solve on them equals a solve on uploaded host tables
    s2 = BatchedOcpSolver(make_ocp(), 70, np.stack([p.s_ref for p in plans]), np.stack([p.kappa_ref for p in plans]), track_id=np.arange(70) % len(names))
    x0 = np.zeros((70, 8))
    for t, p in enumerate(plans):
        sel = np.arange(70) % len(names) == t
        x0[sel] = sample_x0(p, int(sel.sum()), seed=100 + t)
    for sv in (s, s2):
        sv.set_x0(x0); sv.init_guess()
        yref = np.zeros((70, 40, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(40)[None] / 40
        yref_e = np.zeros((70, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
        sv.set_yref(yref); sv.set_yref_e(yref_e); sv.set_multipliers(None, None)
    st, st2 = s.solve(), s2.solve()
    assert np.mean(st == st2) >= 0.95           # tables agree to 1e-10: an instance at the edge of failure may fall either way
    ok = (st == 0) & (st2 == 0)
    assert ok.mean() > 0.8
    assert np.max(np.abs(s.get_u()[ok] - s2.get_u()[ok]) / np.maximum(1.0, np.abs(s2.get_u()[ok]))) < 1e-6
    # misuse
    with pytest.raises(Exception, match="max_seg|segments"):
        s.build_tracks([c[0][:1] for c in cs], [c[1][:1] for c in cs])
    s.free(); s2.free()


def test_device_spline_fit_matches_the_independent_restatement_and_feeds_the_tables():
    """SURVEY.md section 8f N1, the remainder: fit_spline (python/motion_planning.py:28-124) on the device -- `ihm2mpc_fit_tracks`, the KKT system
    of the reference's programme by sparse Gaussian elimination, one workgroup per track -- against the oracle's null-space fit on all seven
    tracks (1e-7 of the track's extent: two solution methods of a programme with a 1e-10 regulariser), its KKT residuals, and then the whole
    pipeline on the device: centre line -> coefficients -> tables, against the oracle's pipeline."""
    from ihm2_amd import track as T
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import track_np as OT

    names = T.TRACK_NAMES
    geos = [T.load_track_geometry_data(n) for n in names]
    nk = 3 * T.NUMBER_SPLINE_INTERVALS
    dummy = np.tile(np.linspace(0.0, 1.0, nk), (len(names), 1))
    s = BatchedOcpSolver(make_ocp(), 14, dummy, np.zeros_like(dummy), track_id=np.arange(14) % len(names))
    cX, cY = s.fit_tracks([g.center_line for g in geos], curv_weight=2.0)
    for t, g in enumerate(geos):
        oX, oY = OT.fit_spline_nullspace(g.center_line, 2.0)
        ext = np.ptp(g.center_line, axis=0).max()
        assert cX[t].shape == oX.shape
        assert np.max(np.abs(cX[t] - oX)) < 1e-7 * ext and np.max(np.abs(cY[t] - oY)) < 1e-7 * ext, names[t]
        feas, stat = OT.kkt_residuals(g.center_line, cX[t], cY[t], 2.0)
        assert feas < 1e-9 * ext and stat < 1e-8 * ext, (names[t], feas, stat)
    s.build_tracks(cX, cY)
    s_ref, kappa, X, Y, phi = s.get_tracks()
    for t, g in enumerate(geos):
        mp = OT.motion_plan(g.center_line, g.track_widths)
        L = mp["lap_length"]
        assert np.max(np.abs(s_ref[t] - mp["s_ref"])) < 1e-7 * L and np.max(np.abs(X[t] - mp["X_ref"])) < 1e-7 * L and np.max(np.abs(Y[t] - mp["Y_ref"])) < 1e-7 * L
        assert np.max(np.abs(kappa[t] - mp["kappa_ref"])) < 1e-6 * max(1.0, np.max(np.abs(mp["kappa_ref"])))
    # another weight, a circle (closed form), and misuse
    th = 2 * np.pi * np.arange(40) / 40
    circ = np.column_stack((20 * np.cos(th), 20 * np.sin(th)))
    s1 = BatchedOcpSolver(make_ocp(), 2, dummy[:1], np.zeros_like(dummy[:1]))
    c1X, c1Y = s1.fit_tracks([circ], curv_weight=0.5)
    o1X, o1Y = OT.fit_spline_nullspace(circ, 0.5)
    assert np.max(np.abs(c1X[0] - o1X)) < 1e-8 * 40 and np.max(np.abs(c1Y[0] - o1Y)) < 1e-8 * 40
    assert np.max(np.abs(np.hypot(c1X[0][:, 0], c1Y[0][:, 0]) - 20.0)) < 0.05          # the smoothed points stay on the circle
    with pytest.raises(Exception, match="max_pts|points"):
        s1.fit_tracks([circ[:2]])
    s.free(); s1.free()
