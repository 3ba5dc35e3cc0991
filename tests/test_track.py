"""Track preprocessing (python/motion_planning.py restated in ihm2_amd/track.py)."""
import numpy as np
import pytest

from ihm2_amd import track as T


def test_all_reference_tracks_load():
    sizes = {"acceleration": 37, "fsds_competition_1": 87, "fsds_competition_2": 117, "fsds_competition_3": 92,
             "fsds_default": 98, "short_skidpad": 80, "skidpad": 140}
    for name, n in sizes.items():
        g = T.load_track_geometry_data(name)
        assert g.center_line.shape == (n, 2) and g.track_widths.shape == (n, 2)
    with pytest.raises(FileNotFoundError):
        T.load_track_geometry_data("nope")


def test_spline_fit_of_a_circle():
    R, n = 25.0, 60
    th = np.linspace(0, 2 * np.pi, n, endpoint=False)
    path = np.column_stack((R * np.cos(th), R * np.sin(th)))
    cX, cY = T.fit_spline(path, curv_weight=2.0)
    # C0/C1/C2 continuity at the junctions (C1/C2 scaled by the chord ratio = 1 here)
    for c in (cX, cY):
        nxt = np.roll(c, -1, axis=0)
        np.testing.assert_allclose(c.sum(1), nxt[:, 0], atol=1e-9)
        np.testing.assert_allclose(c[:, 1] + 2 * c[:, 2] + 3 * c[:, 3], nxt[:, 1], atol=1e-9)
        np.testing.assert_allclose(2 * c[:, 2] + 6 * c[:, 3], 2 * nxt[:, 2], atol=1e-9)
    ds = T.compute_spline_interval_lengths(cX, cY)
    assert ds.sum() == pytest.approx(2 * np.pi * R, rel=2e-3)
    X, Y, idx, t, s = T.uniformly_sample_spline(cX, cY, ds, 200)
    kap = T.get_curvature(cX, cY, idx, t)
    np.testing.assert_allclose(kap, 1.0 / R, rtol=2e-2)
    assert s[0] == 0.0 and np.allclose(np.diff(s), ds.sum() / 200)


def test_motion_plan_of_reference_track():
    mp = T.offline_motion_plan("fsds_competition_1")
    assert mp.s_ref.shape == (500,) and mp.kappa_ref.shape == (500,)
    # closed polyline ~339.8 m, 500 samples at ds ~0.68 m, min half width 1.675 m
    assert 338.0 < mp.lap_length < 342.0
    assert np.diff(mp.s_ref).mean() == pytest.approx(0.68, abs=0.01)
    assert mp.right_widths[0] == pytest.approx(1.675, abs=1e-3)
    assert np.abs(mp.kappa_ref).max() < 0.25
    tp = T.triple_motion_plan_ref(mp)
    assert tp.s_ref.shape == (1500,) and np.all(np.diff(tp.s_ref) > 0)
    np.testing.assert_allclose(tp.s_ref[500:1000], mp.s_ref)
    np.testing.assert_allclose(tp.s_ref[:500], mp.s_ref - mp.lap_length)
    np.testing.assert_array_equal(tp.kappa_ref[1000:], mp.kappa_ref)
    assert tp.p.shape == (3000,)


def test_track_file_format(tmp_path):
    out = tmp_path / "t" / "fsds_competition_1.csv"
    T.generate_track_data_file("fsds_competition_1", str(out))
    lines = out.read_text().splitlines()
    assert lines[0] == "s_ref,X_ref,Y_ref,phi_ref,kappa_ref,right_width,left_width"
    assert len(lines) == 501 and len(lines[1].split(",")) == 7
    assert all(len(v.split(".")[1]) == 6 for v in lines[1].split(","))
