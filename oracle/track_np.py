"""CPU ORACLE (test infrastructure, NOT the product): the reference's offline motion planner restated independently of
``ihm2_amd/track.py`` -- a second implementation the product's host planner AND the device track kernels (``ihm2mpc_build_tracks``)
are checked against.  PARITY UNPINNED like the rest of ``oracle/``: ``qpsolvers`` / ``proxsuite`` / ``track_database`` are not
installed here, the reference holds no track fixtures beyond the centre-line CSVs; pinned by closed forms (circle), by
``scipy.interpolate.CubicSpline(bc_type="periodic")`` in the limit ``curv_weight -> 0`` and by the programme's own KKT conditions
(``tests/test_oracle_track.py``).

Follows (reference file:line, relative to /root/reference):
  fit_spline                        python/motion_planning.py:28-124   (equality-constrained least squares; the reference hands it to
                                    qpsolvers/proxqp -- here: elimination of the constraints by a null-space basis, then an
                                    unconstrained solve; the product solves the dense KKT system instead)
  compute_spline_interval_lengths   :139-177   uniformly_sample_spline :180-232   get_heading :235-262   get_curvature :265-289
  offline_motion_plan               :345-399   triple_motion_plan_ref  :402-428
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import null_space


def continuity_matrix(path: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """``A`` of ``python/motion_planning.py:59-87`` (3N x 4N): value, first and second derivative of segment i at t = 1 equal those of
    segment i + 1 (cyclic) at t = 0, the derivatives scaled by the chord ratio rho_i = ds_i / ds_{i+1} and its square; and the chords."""
    n = path.shape[0]
    ds = np.linalg.norm(np.roll(path, -1, axis=0) - path, axis=1)        # ds_i = |p_{i+1} - p_i|, closing chord last (:54-55)
    rho = ds / np.roll(ds, -1)                                           # :56-58
    A = np.zeros((3 * n, 4 * n))
    for i in range(n):
        nxt = (i + 1) % n
        A[3 * i, 4 * i:4 * i + 4] = [1, 1, 1, 1]
        A[3 * i + 1, 4 * i:4 * i + 4] = [0, 1, 2, 3]
        A[3 * i + 2, 4 * i:4 * i + 4] = [0, 0, 2, 6]
        A[3 * i, 4 * nxt] += -1.0
        A[3 * i + 1, 4 * nxt + 1] += -rho[i]
        A[3 * i + 2, 4 * nxt + 2] += -2.0 * rho[i] ** 2
    return A, ds


def spline_cost(path: np.ndarray, ds: np.ndarray, curv_weight: float) -> tuple[np.ndarray, np.ndarray]:
    """``P = B'B + w C'C + 1e-10 I`` and ``q = -B' path`` (``python/motion_planning.py:88-101``)."""
    n = path.shape[0]
    B = np.zeros((n, 4 * n)); C = np.zeros((n, 4 * n))
    for i in range(n):
        B[i, 4 * i] = 1.0
        C[i, 4 * i + 2] = 2.0 / ds[i] ** 2
        C[i, 4 * i + 3] = 6.0 / ds[i] ** 2
    return B.T @ B + curv_weight * C.T @ C + 1e-10 * np.eye(4 * n), -B.T @ path


def fit_spline_nullspace(path: np.ndarray, curv_weight: float = 1.0) -> tuple[np.ndarray, np.ndarray]:
    """min 1/2 p'P p + q'p  s.t.  A p = 0, for x and y: p = Z y with Z an orthonormal basis of ker A, (Z'PZ) y = -Z'q."""
    path = np.asarray(path, dtype=np.float64)
    A, ds = continuity_matrix(path)
    P, q = spline_cost(path, ds, curv_weight)
    Z = null_space(A)
    assert Z.shape[1] == path.shape[0], "ker A has one dimension per path point"
    y = np.linalg.solve(Z.T @ P @ Z, -Z.T @ q)
    p = Z @ y
    n = path.shape[0]
    return p[:, 0].reshape(n, 4), p[:, 1].reshape(n, 4)


def kkt_residuals(path: np.ndarray, cX: np.ndarray, cY: np.ndarray, curv_weight: float) -> tuple[float, float]:
    """(max |A p|, max |Z'(P p + q)|) of a candidate solution: feasibility and stationarity on the feasible subspace."""
    A, ds = continuity_matrix(np.asarray(path, dtype=np.float64))
    P, q = spline_cost(np.asarray(path, dtype=np.float64), ds, curv_weight)
    p = np.column_stack((cX.reshape(-1), cY.reshape(-1)))
    Z = null_space(A)
    return float(np.max(np.abs(A @ p))), float(np.max(np.abs(Z.T @ (P @ p + q))))


def _poly(c, t, der=0):
    if der == 0:
        return c[..., 0] + c[..., 1] * t + c[..., 2] * t ** 2 + c[..., 3] * t ** 3
    if der == 1:
        return c[..., 1] + 2 * c[..., 2] * t + 3 * c[..., 3] * t ** 2
    return 2 * c[..., 2] + 6 * c[..., 3] * t


def interval_lengths(cX: np.ndarray, cY: np.ndarray, no_interp_points: int = 100) -> np.ndarray:
    """Polyline length of every segment on ``no_interp_points`` points (``python/motion_planning.py:139-177``)."""
    out = np.zeros(cX.shape[0])
    t = np.linspace(0.0, 1.0, no_interp_points)
    for i in range(cX.shape[0]):
        x, y = _poly(cX[i], t), _poly(cY[i], t)
        out[i] = np.sum(np.sqrt(np.diff(x) ** 2 + np.diff(y) ** 2))
    return out


def motion_plan(center_line: np.ndarray, widths: np.ndarray, n_samples: int = 500, l_R: float = 0.7853, curv_weight: float = 2.0,
                coeffs: tuple[np.ndarray, np.ndarray] | None = None) -> dict:
    """``offline_motion_plan`` + ``triple_motion_plan_ref`` (``python/motion_planning.py:345-428``): dict of s_ref, X_ref, Y_ref, phi_ref,
    kappa_ref (3 n_samples each), right / left widths, lap_length.  ``coeffs``: spline coefficients to start from (default: own fit)."""
    cX, cY = fit_spline_nullspace(center_line, curv_weight) if coeffs is None else coeffs
    ds = interval_lengths(cX, cY)
    s_end = np.cumsum(ds)
    s = np.linspace(0.0, s_end[-1], n_samples, endpoint=False)                      # :203
    X = np.zeros(n_samples); Y = np.zeros(n_samples); phi = np.zeros(n_samples); kap = np.zeros(n_samples)
    for m in range(n_samples):
        i = int(np.argmax(s[m] < s_end))                                            # :206
        t = (s[m] - (s_end[i - 1] if i > 0 else 0.0)) / ds[i]                       # :213-216
        X[m], Y[m] = _poly(cX[i], t), _poly(cY[i], t)
        xd, yd, xdd, ydd = _poly(cX[i], t, 1), _poly(cY[i], t, 1), _poly(cX[i], t, 2), _poly(cY[i], t, 2)
        kap[m] = (xd * ydd - yd * xdd) / (xd ** 2 + yd ** 2) ** 1.5                 # :287
        phi[m] = np.arctan2(yd, xd) - np.arcsin(l_R * kap[m])                       # :260, :380
    L = s[-1] + np.hypot(X[-1] - X[0], Y[-1] - Y[0])                                # :386
    rep = lambda a: np.hstack((a, a, a))                                            # noqa: E731  (:402-428)
    return dict(s_ref=np.hstack((s - L, s, s + L)), X_ref=rep(X), Y_ref=rep(Y), phi_ref=rep(phi), kappa_ref=rep(kap),
                right_width=float(np.min(widths[:, 0])), left_width=float(np.min(widths[:, 1])), lap_length=float(L))
