"""SURVEY.md section 8f N1, pinned independently of the product: ``oracle/track_np.py`` restates python/motion_planning.py:28-428 with another
solution method (null-space elimination instead of the product's dense KKT solve) and is itself pinned by closed forms, by scipy's periodic
cubic spline in the limit curv_weight -> 0 and by the programme's KKT conditions.  The product's host planner (``ihm2_amd/track.py``) is then
checked against it on every track of data/ -- and so are the device tables (tests/test_gpu_tracks.py)."""
import numpy as np
import pytest

from ihm2_amd import track as T
from oracle import track_np as O


def test_nullspace_fit_satisfies_the_kkt_conditions_of_the_programme():
    geo = T.load_track_geometry_data("fsds_competition_1")
    cX, cY = O.fit_spline_nullspace(geo.center_line, curv_weight=2.0)
    feas, stat = O.kkt_residuals(geo.center_line, cX, cY, 2.0)
    assert feas < 1e-9 and stat < 1e-9
    # interpolation error of the smoothed spline stays in the centimetre range (python/motion_planning.py:108-110 reports it)
    assert np.max(np.hypot(cX[:, 0] - geo.center_line[:, 0], cY[:, 0] - geo.center_line[:, 1])) < 0.25


def test_limit_of_no_smoothing_is_the_periodic_interpolating_cubic_spline():
    """curv_weight -> 0: the programme interpolates the points with C2 continuity in the chord-length parameter -- scipy's
    CubicSpline(bc_type="periodic") over the cumulative chord length is the same curve."""
    from scipy.interpolate import CubicSpline

    rng = np.random.default_rng(4)
    th = np.sort(rng.uniform(0, 2 * np.pi, 24))
    path = np.column_stack((30 * np.cos(th) + 3 * np.cos(3 * th), 20 * np.sin(th)))
    cX, cY = O.fit_spline_nullspace(path, curv_weight=1e-9)
    ds = np.linalg.norm(np.roll(path, -1, axis=0) - path, axis=1)
    u = np.concatenate(([0.0], np.cumsum(ds)))
    cs = CubicSpline(u, np.vstack((path, path[:1])), bc_type="periodic")
    for i in range(24):
        t = np.linspace(0, 1, 7)
        got = np.column_stack((O._poly(cX[i], t), O._poly(cY[i], t)))
        np.testing.assert_allclose(got, cs(u[i] + t * ds[i]), atol=2e-5)          # 1e-9 of smoothing and of the 1e-10 regulariser are left


def test_circle_has_constant_curvature_and_the_right_length():
    R, n = 25.0, 60
    th = 2 * np.pi * np.arange(n) / n
    path = np.column_stack((R * np.cos(th), R * np.sin(th)))
    mp = O.motion_plan(path, np.full((n, 2), 1.5), n_samples=200)
    assert abs(mp["lap_length"] - 2 * np.pi * R) < 2e-3 * R
    np.testing.assert_allclose(mp["kappa_ref"], 1.0 / R, rtol=2e-3)
    assert np.all(np.diff(mp["s_ref"]) > 0) and mp["s_ref"].size == 600


@pytest.mark.parametrize("name", T.TRACK_NAMES)
def test_product_planner_matches_the_independent_restatement(name):
    """ihm2_amd/track.py (dense KKT fit, vectorised sampling) against oracle/track_np.py (null-space fit, point-by-point sampling) on every
    track of data/, the three open ones included: spline coefficients to 1e-7 of the track's extent, tables to 1e-7 of a lap / of the largest curvature."""
    geo = T.load_track_geometry_data(name)
    cX, cY = T.fit_spline(geo.center_line, curv_weight=2.0)
    oX, oY = O.fit_spline_nullspace(geo.center_line, curv_weight=2.0)
    ext = np.ptp(geo.center_line, axis=0).max()
    assert np.max(np.abs(cX - oX)) < 1e-7 * ext and np.max(np.abs(cY - oY)) < 1e-7 * ext
    plan = T.track_table(name)
    mp = O.motion_plan(geo.center_line, geo.track_widths)
    L = mp["lap_length"]
    assert abs(plan.lap_length - L) < 1e-7 * L
    assert np.max(np.abs(plan.s_ref - mp["s_ref"])) < 1e-7 * L
    assert np.max(np.abs(plan.X_ref - mp["X_ref"])) < 1e-7 * L and np.max(np.abs(plan.Y_ref - mp["Y_ref"])) < 1e-7 * L
    assert np.max(np.abs(plan.kappa_ref - mp["kappa_ref"])) < 1e-6 * max(1.0, np.max(np.abs(mp["kappa_ref"])))
    assert np.max(np.abs(np.angle(np.exp(1j * (plan.phi_ref - mp["phi_ref"]))))) < 1e-6
    assert plan.right_widths[0] == mp["right_width"] and plan.left_widths[0] == mp["left_width"]
