"""Track preprocessing: centre line -> path parameter table ``p = [s_ref; kappa_ref]``.

Host-side NumPy restatement of the reference's offline motion planner
(``python/motion_planning.py``): closed cubic-spline fit (``fit_spline`` :28-124, an
equality-constrained least-squares problem, solved here by one dense KKT solve instead of
``qpsolvers/proxqp``), polyline arc length per segment (:139-177), uniform arc-length resampling
(:180-232), heading / curvature (:235-289), ``offline_motion_plan`` (:345-399) and the 3-lap tiling
``triple_motion_plan_ref`` (:402-428).  ``NUMBER_SPLINE_INTERVALS = 500`` (:25).

The table feeds the curvature interpolant of the vehicle models (``python/models.py:290-295``).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

from .constants import l_R

NUMBER_SPLINE_INTERVALS = 500
DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
TRACK_NAMES = (
    "acceleration",
    "fsds_competition_1",
    "fsds_competition_2",
    "fsds_competition_3",
    "fsds_default",
    "short_skidpad",
    "skidpad",
)
# tracks whose centre line closes on itself; the spline fit assumes closure
CLOSED_TRACK_NAMES = ("fsds_competition_1", "fsds_competition_2", "fsds_competition_3", "fsds_default")


@dataclass
class TrackGeometryData:
    """Name kept from ``new_python/track_data.py:5-10``."""

    X_center_line: np.ndarray
    Y_center_line: np.ndarray
    right_width: np.ndarray
    left_width: np.ndarray

    @property
    def center_line(self) -> np.ndarray:
        return np.column_stack((self.X_center_line, self.Y_center_line))

    @property
    def track_widths(self) -> np.ndarray:
        return np.column_stack((self.right_width, self.left_width))


def load_track_geometry_data(track_name: str) -> TrackGeometryData:
    """``new_python/track_data.py:26-27`` (a stub there); reads the build's own fixture CSV."""
    path = os.path.join(DATA_DIR, track_name + ".csv")
    if not os.path.exists(path):
        raise FileNotFoundError(f"unknown track {track_name!r}; available: {TRACK_NAMES}")
    arr = np.loadtxt(path, delimiter=",", skiprows=1)
    return TrackGeometryData(arr[:, 0].copy(), arr[:, 1].copy(), arr[:, 2].copy(), arr[:, 3].copy())


@dataclass
class MotionPlan:
    """Same fields as ``python/motion_planning.py:292-301``."""

    s_ref: np.ndarray
    X_ref: np.ndarray
    Y_ref: np.ndarray
    phi_ref: np.ndarray
    kappa_ref: np.ndarray
    right_widths: np.ndarray
    left_widths: np.ndarray
    lap_length: float

    @property
    def p(self) -> np.ndarray:
        """Model parameter vector ``[s_ref; kappa_ref]`` (``python/main.py:249``)."""
        return np.append(self.s_ref, self.kappa_ref)


def fit_spline(path: np.ndarray, curv_weight: float = 1.0) -> tuple[np.ndarray, np.ndarray]:
    """Closed cubic spline through (approximately) the path points.

    Segment i is ``c0 + c1 t + c2 t^2 + c3 t^3`` on t in [0,1].  Minimises
    ``sum_i |c0_i - path_i|^2 + curv_weight * sum_i ((2 c2_i + 6 c3_i) / ds_i^2)^2 + 1e-10 |c|^2``
    subject to C0/C1/C2 continuity between consecutive segments (C1/C2 scaled by the chord-length
    ratio), exactly the program of ``python/motion_planning.py:53-113``.
    Returns two (N,4) coefficient arrays (x and y).
    """
    path = np.asarray(path, dtype=np.float64)
    if path.ndim != 2 or path.shape[1] != 2:
        raise ValueError(f"path must have shape (N,2), got {path.shape}")
    n = path.shape[0]
    nxt = np.roll(np.arange(n), -1)
    chord = np.linalg.norm(path[nxt] - path, axis=1)          # ds_i: point i -> point i+1 (closing)
    rho = chord / chord[nxt]
    nv = 4 * n
    # continuity constraints E c = 0, 3 rows per junction
    E = np.zeros((3 * n, nv))
    rows = 3 * np.arange(n)
    for i in range(n):
        j = nxt[i]
        E[rows[i], 4 * i:4 * i + 4] = (1.0, 1.0, 1.0, 1.0)
        E[rows[i], 4 * j] -= 1.0
        E[rows[i] + 1, 4 * i:4 * i + 4] = (0.0, 1.0, 2.0, 3.0)
        E[rows[i] + 1, 4 * j + 1] -= rho[i]
        E[rows[i] + 2, 4 * i:4 * i + 4] = (0.0, 0.0, 2.0, 6.0)
        E[rows[i] + 2, 4 * j + 2] -= 2.0 * rho[i] ** 2
    # quadratic cost
    Q = 1e-10 * np.eye(nv)
    lin = np.zeros((nv, 2))
    for i in range(n):
        Q[4 * i, 4 * i] += 1.0
        lin[4 * i] = -path[i]
        w = np.array([2.0, 6.0]) / chord[i] ** 2
        Q[np.ix_([4 * i + 2, 4 * i + 3], [4 * i + 2, 4 * i + 3])] += curv_weight * np.outer(w, w)
    # KKT system, both coordinates at once
    K = np.block([[Q, E.T], [E, np.zeros((3 * n, 3 * n))]])
    rhs = np.vstack((-lin, np.zeros((3 * n, 2))))
    sol = np.linalg.solve(K, rhs)
    return sol[:nv, 0].reshape(n, 4), sol[:nv, 1].reshape(n, 4)


def _eval_poly(c: np.ndarray, idx: np.ndarray, t: np.ndarray, der: int = 0) -> np.ndarray:
    c0, c1, c2, c3 = (c[idx, k] for k in range(4))
    if der == 0:
        return c0 + t * (c1 + t * (c2 + t * c3))
    if der == 1:
        return c1 + t * (2.0 * c2 + 3.0 * c3 * t)
    return 2.0 * c2 + 6.0 * c3 * t


def compute_spline_interval_lengths(coeffs_X: np.ndarray, coeffs_Y: np.ndarray, no_interp_points: int = 100) -> np.ndarray:
    """Polyline length of each segment on ``no_interp_points`` samples (``motion_planning.py:139-177``)."""
    n = coeffs_X.shape[0]
    t = np.linspace(0.0, 1.0, no_interp_points)
    idx = np.repeat(np.arange(n), no_interp_points)
    tt = np.tile(t, n)
    X = _eval_poly(coeffs_X, idx, tt).reshape(n, no_interp_points)
    Y = _eval_poly(coeffs_Y, idx, tt).reshape(n, no_interp_points)
    return np.sum(np.hypot(np.diff(X, axis=1), np.diff(Y, axis=1)), axis=1)


def uniformly_sample_spline(coeffs_X, coeffs_Y, delta_s, n_samples):
    """``n_samples`` points equidistant in arc length, first = start of segment 0, endpoint excluded
    (``motion_planning.py:180-232``)."""
    s_end = np.cumsum(delta_s)
    s_interp = np.linspace(0.0, s_end[-1], n_samples, endpoint=False)
    idx = np.argmax(s_interp[:, None] < s_end[None, :], axis=1)
    s_start = np.concatenate(([0.0], s_end[:-1]))
    t = (s_interp - s_start[idx]) / delta_s[idx]
    return _eval_poly(coeffs_X, idx, t), _eval_poly(coeffs_Y, idx, t), idx, t, s_interp


def get_heading(coeffs_X, coeffs_Y, idx, t) -> np.ndarray:
    return np.arctan2(_eval_poly(coeffs_Y, idx, t, 1), _eval_poly(coeffs_X, idx, t, 1))


def get_curvature(coeffs_X, coeffs_Y, idx, t) -> np.ndarray:
    xd, yd = _eval_poly(coeffs_X, idx, t, 1), _eval_poly(coeffs_Y, idx, t, 1)
    xdd, ydd = _eval_poly(coeffs_X, idx, t, 2), _eval_poly(coeffs_Y, idx, t, 2)
    return (xd * ydd - yd * xdd) / np.power(xd * xd + yd * yd, 1.5)


def offline_motion_plan(track: str | TrackGeometryData, n_samples: int = NUMBER_SPLINE_INTERVALS) -> MotionPlan:
    """``python/motion_planning.py:345-399`` (curv_weight=2.0; heading offset by ``-asin(l_R*kappa)``)."""
    if isinstance(track, str):
        track = load_track_geometry_data(track)
    cX, cY = fit_spline(track.center_line, curv_weight=2.0)
    delta_s = compute_spline_interval_lengths(cX, cY)
    X_ref, Y_ref, idx, t, s_ref = uniformly_sample_spline(cX, cY, delta_s, n_samples)
    kappa_ref = get_curvature(cX, cY, idx, t)
    phi_ref = get_heading(cX, cY, idx, t) - np.arcsin(l_R * kappa_ref)
    right = np.tile(np.min(track.right_width), n_samples)
    left = np.tile(np.min(track.left_width), n_samples)
    lap_length = float(s_ref[-1] + np.hypot(X_ref[-1] - X_ref[0], Y_ref[-1] - Y_ref[0]))
    return MotionPlan(s_ref, X_ref, Y_ref, phi_ref, kappa_ref, right, left, lap_length)


def triple_motion_plan_ref(mp: MotionPlan) -> MotionPlan:
    """Three laps side by side, s shifted by -L, 0, +L (``python/motion_planning.py:402-428``)."""
    L = mp.lap_length
    rep = lambda a: np.hstack((a, a, a))  # noqa: E731
    return MotionPlan(
        np.hstack((mp.s_ref - L, mp.s_ref, mp.s_ref + L)), rep(mp.X_ref), rep(mp.Y_ref), rep(mp.phi_ref),
        rep(mp.kappa_ref), rep(mp.right_widths), rep(mp.left_widths), L,
    )


def generate_track_data_file(track_name: str, outfile: str) -> None:
    """CSV ``s_ref,X_ref,Y_ref,phi_ref,kappa_ref,right_width,left_width`` with ``%.6f``
    (``python/motion_planning.py:431-456``)."""
    mp = offline_motion_plan(track_name)
    os.makedirs(os.path.dirname(os.path.abspath(outfile)), exist_ok=True)
    cols = np.column_stack((mp.s_ref, mp.X_ref, mp.Y_ref, mp.phi_ref, mp.kappa_ref, mp.right_widths, mp.left_widths))
    np.savetxt(outfile, cols, delimiter=",", fmt="%.6f", comments="",
               header="s_ref,X_ref,Y_ref,phi_ref,kappa_ref,right_width,left_width")


_PLAN_CACHE: dict[tuple[str, int], MotionPlan] = {}


def track_table(track_name: str, n_samples: int = NUMBER_SPLINE_INTERVALS) -> MotionPlan:
    """Tripled motion plan of a named track (cached): ``s_ref``/``kappa_ref`` have 3*n_samples knots."""
    key = (track_name, n_samples)
    if key not in _PLAN_CACHE:
        _PLAN_CACHE[key] = triple_motion_plan_ref(offline_motion_plan(track_name, n_samples))
    return _PLAN_CACHE[key]
