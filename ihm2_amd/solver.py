"""Batched NMPC solver on MI355X behind the reference's ``AcadosOcpSolver`` call surface.

Reference call sites this mirrors (tudoroancea/ihm2):
  * construction ``AcadosOcpSolver(ocp, json_file=..., verbose=False)`` -- ``python/mpc.py:104-113``
  * ``solver.set(stage, field, value)`` with fields ``"p","lbx","ubx","yref","x","u"``
    -- ``python/main.py:252,299-322``
  * ``solver.cost_set(stage, "W", W[, api="new"])`` -- ``python/main.py:253-295``, ``dpc/main.py:263-281``
  * ``solver.constraints_set(stage, "lbx"/"ubx"/"C"..., v)`` -- ``dpc/main.py:226-227,259-261``
  * ``status = solver.solve()`` -- ``python/main.py:325``; status codes ``dpc/main.py:287-293``
  * ``solver.get(stage, "x"/"u")`` -- ``python/main.py:331-334``

Two layers: :class:`BatchedOcpSolver` owns one ``ihm2mpc`` handle (B instances on one GPU) and speaks
``(B, ...)`` C-contiguous float64 arrays; :class:`AcadosOcpSolver` is the per-instance shim
(``view = batch[i]``, or a batch of one when built from an OCP like the reference does).
Host code is NumPy + ctypes only; all arithmetic runs in ``libihm2mpc.so`` (HIP, gfx950).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .constants import NG, NLAM, NU, NX, NY
from .ocp import AcadosOcp, AcadosOcpOptions, OcpData

_SOLVER_TYPE = {"SQP_RTI": 0, "SQP": 1}
_GLOBALIZATION = {"FIXED_STEP": 0, "MERIT_BACKTRACKING": 1}


def _f64(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {a.shape}")
    return a


def _ptr(a):
    return a.ctypes.data_as(_lib.c_double_p)


class BatchedOcpSolver:
    """B independent instances of the bicycle NMPC on one MI355X."""

    def __init__(self, ocp: AcadosOcp, batch_size: int, s_ref, kappa_ref, track_id=None, device: int = 0, track_widths=None):
        self.lib = _lib.load()
        self.ocp = ocp
        self.data: OcpData = ocp.flatten()
        d = self.data
        self.B, self.N, self.device = int(batch_size), d.N, int(device)
        s_ref = np.atleast_2d(_f64(s_ref)); kappa_ref = np.atleast_2d(_f64(kappa_ref))
        if s_ref.shape != kappa_ref.shape:
            raise ValueError("s_ref and kappa_ref must have the same shape (ntracks, nknots)")
        self.ntracks, self.nknots = s_ref.shape
        cfg = _lib.Config(
            batch=self.B, N=d.N, M=d.M, model=d.model, ntracks=self.ntracks, nknots=self.nknots, device=device,
            nlp_solver_type=_SOLVER_TYPE[d.nlp_solver_type], nlp_solver_max_iter=d.nlp_solver_max_iter,
            ipm_iter_max=d.ipm_iter_max, dt=d.dt, cost_scale_stage=d.cost_scale_stage, ipm_tol=d.ipm_tol,
            ipm_mu0=d.ipm_mu0, ipm_tau0=d.ipm_tau0, nlp_tol=d.nlp_tol, integrator_type=d.integrator, sim_integrator_type=d.sim_integrator)
        self._h = C.c_void_p()
        _lib.check(self.lib.ihm2mpc_create(C.byref(cfg), C.byref(self._h)))
        self._s_ref, self._kappa_ref = s_ref.copy(), kappa_ref.copy()
        self._track_widths = None if track_widths is None else np.atleast_2d(_f64(track_widths))
        self._push_tracks()
        self.set_track_id(np.zeros(self.B, dtype=np.int32) if track_id is None else track_id)
        self._push_weights()
        self._push_bounds()
        if d.nlp_solver_type == "SQP":
            self.set_sqp_options(d.globalization, d.alpha_min, d.alpha_reduction, d.eps_sufficient_descent, d.use_sufficient_descent,
                                 d.full_step_dual, d.sqp_tol)

    # ---- lifetime ----
    def free(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ihm2mpc_synchronize(self._h)
            for ptr in getattr(self, "_pinned", []):
                self.lib.ihm2mpc_host_free(ptr)
            self._pinned = []
            self.lib.ihm2mpc_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __len__(self):
        return self.B

    def __getitem__(self, i: int) -> "AcadosOcpSolver":
        if not -self.B <= i < self.B:
            raise IndexError(i)
        return AcadosOcpSolver._view(self, i % self.B)

    # ---- shared problem data ----
    def _push_tracks(self):
        _lib.check(self.lib.ihm2mpc_set_tracks(self._h, _ptr(self._s_ref), _ptr(self._kappa_ref)))

    def _push_weights(self):
        d = self.data
        d.W = _f64(d.W, (self.N, NY, NY), "W"); d.W_e = _f64(d.W_e, (NX, NX), "W_e")
        _lib.check(self.lib.ihm2mpc_set_weights(self._h, _ptr(d.W), _ptr(d.W_e)))

    def _push_bounds(self):
        d = self.data
        arrs = [_f64(getattr(d, n)) for n in ("lbx", "ubx", "lbu", "ubu", "C", "D", "lg", "ug")]
        _lib.check(self.lib.ihm2mpc_set_bounds(self._h, *[_ptr(a) for a in arrs]))
        self._push_path()
        self._push_soft()

    def _push_path(self):
        d = self.data
        if not d.path_on:
            _lib.check(self.lib.ihm2mpc_set_path_constraints(self._h, 0, 0.0, 0.0, None, None, None))
            return
        if self._track_widths is None or self._track_widths.shape != (self.ntracks, 2):
            raise ValueError(f"the track rows (model.con_h_expr) need track_widths of shape ({self.ntracks}, 2) = (right, left)")
        lh, uh = _f64(d.lh, (2,), "lh"), _f64(d.uh, (2,), "uh")
        _lib.check(self.lib.ihm2mpc_set_path_constraints(self._h, 1, float(d.car_L), float(d.car_W), _ptr(self._track_widths),
                                                         _ptr(lh), _ptr(uh)))
        self._push_alat()

    def _push_alat(self):
        """The lateral-acceleration row of the kinematic constraint set (``old/generate_acaods_interface.py:198-209``)."""
        d = self.data
        if not getattr(d, "alat_on", 0):
            _lib.check(self.lib.ihm2mpc_set_alat_constraint(self._h, 0, 0.0, 0.0, None, None))
            return
        z = None if d.alat_soft_z is None else _f64(d.alat_soft_z, (2,), "alat_soft_z")
        Z = None if d.alat_soft_Z is None else _f64(d.alat_soft_Z, (2,), "alat_soft_Z")
        _lib.check(self.lib.ihm2mpc_set_alat_constraint(self._h, 1, float(d.alat_lb), float(d.alat_ub), None if z is None else _ptr(z),
                                                        None if Z is None else _ptr(Z)))

    def _push_soft(self):
        d = self.data
        if getattr(d, "soft_Z", None) is None or not np.any(np.asarray(d.soft_Z) >= 0.0):
            _lib.check(self.lib.ihm2mpc_set_soft(self._h, None, None))
            return
        z = _f64(d.soft_z, (self.N + 1, NLAM), "soft_z"); Z = _f64(d.soft_Z, (self.N + 1, NLAM), "soft_Z")
        _lib.check(self.lib.ihm2mpc_set_soft(self._h, _ptr(z), _ptr(Z)))

    def set_soft(self, soft_z=None, soft_Z=None):
        """Soft constraint sides, ``(N+1, 28)`` each (14 lower then 14 upper; ``soft_Z < 0`` = hard); ``None`` = all hard."""
        self.data.soft_z = None if soft_z is None else np.asarray(soft_z, dtype=np.float64)
        self.data.soft_Z = None if soft_Z is None else np.asarray(soft_Z, dtype=np.float64)
        self._push_soft()

    def set_tracks(self, s_ref, kappa_ref):
        self._s_ref = _f64(np.atleast_2d(s_ref), (self.ntracks, self.nknots), "s_ref")
        self._kappa_ref = _f64(np.atleast_2d(kappa_ref), (self.ntracks, self.nknots), "kappa_ref")
        self._push_tracks()

    def set_track_id(self, track_id):
        tid = np.ascontiguousarray(track_id, dtype=np.int32)
        if tid.shape != (self.B,):
            raise ValueError(f"track_id must have shape ({self.B},)")
        self._track_id = tid
        _lib.check(self.lib.ihm2mpc_set_track_id(self._h, tid.ctypes.data_as(_lib.c_int32_p)))

    def set_weights(self, W=None, W_e=None):
        """W: (12,12) for all stages or (N,12,12); W_e: (8,8).  Shared by the batch."""
        if W is not None:
            W = np.asarray(W, dtype=np.float64)
            self.data.W = np.tile(W[None], (self.N, 1, 1)) if W.ndim == 2 else W
        if W_e is not None:
            self.data.W_e = np.asarray(W_e, dtype=np.float64)
        self._push_weights()

    # ---- per-instance data, batched ----
    def set_x0(self, x0):
        _lib.check(self.lib.ihm2mpc_set_x0(self._h, _ptr(_f64(x0, (self.B, NX), "x0"))))

    def set_x(self, x):
        _lib.check(self.lib.ihm2mpc_set_x(self._h, _ptr(_f64(x, (self.B, self.N + 1, NX), "x"))))

    def set_u(self, u):
        _lib.check(self.lib.ihm2mpc_set_u(self._h, _ptr(_f64(u, (self.B, self.N, NU), "u"))))

    def set_yref(self, yref):
        _lib.check(self.lib.ihm2mpc_set_yref(self._h, _ptr(_f64(yref, (self.B, self.N, NY), "yref"))))

    def set_yref_e(self, yref_e):
        _lib.check(self.lib.ihm2mpc_set_yref_e(self._h, _ptr(_f64(yref_e, (self.B, NX), "yref_e"))))

    def set_multipliers(self, pi=None, lam=None):
        pi_a = None if pi is None else _f64(pi, (self.B, self.N + 1, NX), "pi")
        lam_a = None if lam is None else _f64(lam, (self.B, self.N + 1, NLAM), "lam")
        _lib.check(self.lib.ihm2mpc_set_multipliers(self._h, None if pi_a is None else _ptr(pi_a),
                                                    None if lam_a is None else _ptr(lam_a)))

    # ---- the hot path ----
    def init_guess(self, v_ref_scale: float = 1.0):
        _lib.check(self.lib.ihm2mpc_init_guess(self._h, float(v_ref_scale)))

    def reinit_failed(self, v_ref_scale: float = 1.0):
        """Re-roll the warm start of the instances whose last solve failed (status != 0) from their current ``x0``."""
        _lib.check(self.lib.ihm2mpc_reinit_failed(self._h, float(v_ref_scale)))

    def prepare_step(self, s_target: float):
        """Reference ramp + warm-start shift of ``compute_control`` (``python/main.py:303-322``) on device."""
        _lib.check(self.lib.ihm2mpc_prepare_step(self._h, float(s_target)))

    def solve_async(self, n_iter: int = 0):
        _lib.check(self.lib.ihm2mpc_solve(self._h, int(n_iter)))

    def solve(self, n_iter: int = 0) -> np.ndarray:
        """Runs the RTI iteration(s); returns the per-instance acados status codes, int32[B]."""
        self.solve_async(n_iter)
        return self.get_status()

    def compute_control(self, x0, s_target: float):
        """``IHM2Controller.compute_control`` (``python/main.py:297-334``) for the whole batch in one call: returns ``(u0, status)``."""
        u0 = np.empty((self.B, NU)); st = np.empty(self.B, dtype=np.int32)
        _lib.check(self.lib.ihm2mpc_compute_control(self._h, _ptr(_f64(x0, (self.B, NX), "x0")), float(s_target), _ptr(u0),
                                                    st.ctypes.data_as(_lib.c_int32_p)))
        return u0, st

    def reserve_history(self, n_steps: int):
        """Device room for the histories of ``run_steps`` calls of up to ``n_steps`` steps."""
        _lib.check(self.lib.ihm2mpc_reserve_history(self._h, int(n_steps)))

    def run_steps(self, s_target: float, n_steps: int, model: int = 0, M_sim: int = 25, freeze: bool = False, lap_stop: float = np.inf,
                  u0_hist=None, x0_hist=None, status_hist=None, qp_iter_hist=None, wait: bool = True):
        """``n_steps`` control steps (plant + ``compute_control``) in one launch, every instance running ahead on its own
        wavefront (``ihm2mpc_run_steps``).  Histories are written into the given arrays -- ``(n_steps, B, 2)``, ``(n_steps, B, 8)``,
        ``(n_steps, B)`` int32 twice -- or, for ``True``, into fresh ones; returns a dict of those that were asked for."""
        n = int(n_steps)
        out = {}
        def buf(v, shape, dtype, name):
            if v is None or v is False:
                return None
            a = np.empty(shape, dtype=dtype) if v is True else v
            if a.shape != shape or a.dtype != dtype or not a.flags.c_contiguous:
                raise ValueError(f"{name} must be a C-contiguous {np.dtype(dtype).name} array of shape {shape}")
            out[name] = a
            return a
        u = buf(u0_hist, (n, self.B, NU), np.float64, "u0"); x = buf(x0_hist, (n, self.B, NX), np.float64, "x0")
        st = buf(status_hist, (n, self.B), np.int32, "status"); it = buf(qp_iter_hist, (n, self.B), np.int32, "qp_iter")
        _lib.check(self.lib.ihm2mpc_run_steps(
            self._h, int(model), int(M_sim), float(s_target), n, int(bool(freeze)), float(lap_stop),
            None if u is None else _ptr(u), None if x is None else _ptr(x),
            None if st is None else st.ctypes.data_as(_lib.c_int32_p), None if it is None else it.ctypes.data_as(_lib.c_int32_p)))
        if wait:
            self.synchronize()
        return out

    def set_sqp_options(self, globalization="MERIT_BACKTRACKING", alpha_min=0.05, alpha_reduction=0.7, eps_sufficient_descent=1e-4,
                        use_sufficient_descent=False, full_step_dual=False, tol=None):
        """Line search and tolerances of the SQP mode (``python/main.py:230-237``); ``tol`` = (stat, eq, ineq, comp) or a scalar."""
        t = None if tol is None else np.ascontiguousarray(np.broadcast_to(_f64(tol), (4,)))
        _lib.check(self.lib.ihm2mpc_set_sqp_options(self._h, _GLOBALIZATION[globalization], float(alpha_min), float(alpha_reduction),
                                                     float(eps_sufficient_descent), int(bool(use_sufficient_descent)), int(bool(full_step_dual)),
                                                     None if t is None else _ptr(t)))

    def get_sqp_stats(self):
        """QP solves made (B) and last step length (B) of the last SQP-mode solve."""
        it = np.empty(self.B, dtype=np.int32); alpha = np.empty(self.B)
        _lib.check(self.lib.ihm2mpc_get_sqp_stats(self._h, it.ctypes.data_as(_lib.c_int32_p), _ptr(alpha)))
        return {"sqp_iter": it, "alpha": alpha}

    def synchronize(self):
        _lib.check(self.lib.ihm2mpc_synchronize(self._h))

    def linearize(self):
        _lib.check(self.lib.ihm2mpc_linearize(self._h))

    def get_linearization(self):
        A = np.empty((self.B, self.N, NX, NX)); Bm = np.empty((self.B, self.N, NX, NU)); b = np.empty((self.B, self.N, NX))
        _lib.check(self.lib.ihm2mpc_get_linearization(self._h, _ptr(A), _ptr(Bm), _ptr(b)))
        return A, Bm, b

    def _get(self, fn, shape):
        out = np.empty(shape, dtype=np.float64)
        _lib.check(fn(self._h, _ptr(out)))
        return out

    def get_x(self):
        return self._get(self.lib.ihm2mpc_get_x, (self.B, self.N + 1, NX))

    def get_u(self):
        return self._get(self.lib.ihm2mpc_get_u, (self.B, self.N, NU))

    def get_u0(self):
        return self._get(self.lib.ihm2mpc_get_u0, (self.B, NU))

    def get_x0(self):
        return self._get(self.lib.ihm2mpc_get_x0, (self.B, NX))

    def get_residuals(self):
        return self._get(self.lib.ihm2mpc_get_residuals, (self.B, 4))

    def build_tracks(self, coeffs_X, coeffs_Y):
        """Track tables on the device from the centre lines' spline coefficients (``python/motion_planning.py:139-428``): ``coeffs_X``,
        ``coeffs_Y`` are lists of ``(nseg_t, 4)`` arrays, one per track (``track.py::fit_spline``).  Replaces ``set_tracks`` +
        ``set_track_geometry``; ``nknots`` of this solver = 3 x samples per lap."""
        if len(coeffs_X) != self.ntracks or len(coeffs_Y) != self.ntracks:
            raise ValueError(f"{self.ntracks} tracks expected")
        nseg = np.array([len(c) for c in coeffs_X], dtype=np.int32)
        mx = int(nseg.max())
        cX = np.zeros((self.ntracks, mx, 4)); cY = np.zeros((self.ntracks, mx, 4))
        for t in range(self.ntracks):
            cX[t, :nseg[t]] = coeffs_X[t]; cY[t, :nseg[t]] = coeffs_Y[t]
        _lib.check(self.lib.ihm2mpc_build_tracks(self._h, mx, nseg.ctypes.data_as(_lib.c_int32_p), _ptr(cX), _ptr(cY)))
        s_ref = np.empty((self.ntracks, self.nknots)); kappa = np.empty((self.ntracks, self.nknots))
        _lib.check(self.lib.ihm2mpc_get_tracks(self._h, _ptr(s_ref), _ptr(kappa), None, None, None))
        self._s_ref, self._kappa_ref = s_ref, kappa          # the host copies the per-instance shim compares "p" against

    def fit_tracks(self, center_lines, curv_weight: float = 2.0):
        """Closed cubic-spline fit of every track's centre line on the device (``fit_spline``, ``python/motion_planning.py:28-124``; the
        weight 2.0 is ``offline_motion_plan``'s, ``:358``).  ``center_lines``: one ``(npts_t, 2)`` array per track.  Returns two lists of
        ``(npts_t, 4)`` coefficient arrays -- what :meth:`build_tracks` takes."""
        if len(center_lines) != self.ntracks:
            raise ValueError(f"{self.ntracks} tracks expected")
        npts = np.array([len(c) for c in center_lines], dtype=np.int32)
        mx = int(npts.max())
        xy = np.zeros((self.ntracks, mx, 2))
        for t, c in enumerate(center_lines):
            xy[t, :npts[t]] = np.asarray(c, dtype=np.float64)
        cX = np.zeros((self.ntracks, mx, 4)); cY = np.zeros((self.ntracks, mx, 4))
        _lib.check(self.lib.ihm2mpc_fit_tracks(self._h, mx, npts.ctypes.data_as(_lib.c_int32_p), _ptr(xy), float(curv_weight), _ptr(cX), _ptr(cY)))
        return [cX[t, :npts[t]].copy() for t in range(self.ntracks)], [cY[t, :npts[t]].copy() for t in range(self.ntracks)]

    def get_tracks(self):
        """``(s_ref, kappa_ref, X_ref, Y_ref, phi_ref)``, each ``(ntracks, nknots)``, as they are on the device."""
        out = [np.empty((self.ntracks, self.nknots)) for _ in range(5)]
        _lib.check(self.lib.ihm2mpc_get_tracks(self._h, *[_ptr(a) for a in out]))
        return tuple(out)

    def sim_step_dyn10(self, x, u, M_sim: int = 100):
        """Plant step of the 15-state model ``fdyn10`` (python/models.py:609-801): ``x`` (B, 15), ``u`` (B, 5) -> (B, 15)."""
        xn = np.empty((self.B, 15))
        _lib.check(self.lib.ihm2mpc_sim_step_dyn10(self._h, int(M_sim), _ptr(_f64(x, (self.B, 15), "x")), _ptr(_f64(u, (self.B, 5), "u")), _ptr(xn)))
        return xn

    def get_qp_residuals(self):
        """(B, 4) relative KKT residuals of the QP at the returned point (stationarity, dynamics, inequalities, complementarity)."""
        return self._get(self.lib.ihm2mpc_get_qp_residuals, (self.B, 4))

    def get_multipliers(self):
        pi = np.empty((self.B, self.N + 1, NX)); lam = np.empty((self.B, self.N + 1, NLAM))
        _lib.check(self.lib.ihm2mpc_get_multipliers(self._h, _ptr(pi), _ptr(lam)))
        return pi, lam

    def get_alat_multipliers(self):
        """Multipliers and slack values of the lateral-acceleration row, ``(B, N+1, 2)`` each: lower side, upper side."""
        lam = np.empty((self.B, self.N + 1, 2)); slk = np.empty((self.B, self.N + 1, 2))
        _lib.check(self.lib.ihm2mpc_get_alat_multipliers(self._h, _ptr(lam), _ptr(slk)))
        return lam, slk

    def set_alat_multipliers(self, lam=None, slk=None):
        la = None if lam is None else _f64(lam, (self.B, self.N + 1, 2), "lam")
        sa = None if slk is None else _f64(slk, (self.B, self.N + 1, 2), "slk")
        _lib.check(self.lib.ihm2mpc_set_alat_multipliers(self._h, None if la is None else _ptr(la), None if sa is None else _ptr(sa)))

    def set_slacks(self, sl=None):
        """Slack values the next SQP-mode solve starts from, ``(B, N+1, 28)``; ``None`` = zeros."""
        _lib.check(self.lib.ihm2mpc_set_slacks(self._h, None if sl is None else _ptr(_f64(sl, (self.B, self.N + 1, NLAM), "sl"))))

    def get_slacks(self):
        """Slack of each soft constraint side after the last QP, ``(B, N+1, 28)`` (0 for hard sides)."""
        return self._get(self.lib.ihm2mpc_get_slacks, (self.B, self.N + 1, NLAM))

    def get_status(self):
        out = np.empty(self.B, dtype=np.int32)
        _lib.check(self.lib.ihm2mpc_get_status(self._h, out.ctypes.data_as(_lib.c_int32_p)))
        return out

    def get_qp_iter(self):
        out = np.empty(self.B, dtype=np.int32)
        _lib.check(self.lib.ihm2mpc_get_qp_iter(self._h, out.ctypes.data_as(_lib.c_int32_p)))
        return out

    def get_timings(self):
        ms = np.zeros(3)
        _lib.check(self.lib.ihm2mpc_get_timings(self._h, _ptr(ms), 3))
        return {"total_ms": ms[0], "linearize_ms": ms[1], "qp_ms": ms[2]}

    # ---- device-pointer variants (zero copy; dptr = integer device address, instance-major layout) ----
    def set_x0_device(self, dptr: int):
        _lib.check(self.lib.ihm2mpc_set_x0_device(self._h, C.c_void_p(dptr)))

    def get_u0_device(self, dptr: int):
        _lib.check(self.lib.ihm2mpc_get_u0_device(self._h, C.c_void_p(dptr)))

    def get_x_device(self, dptr: int):
        _lib.check(self.lib.ihm2mpc_get_x_device(self._h, C.c_void_p(dptr)))

    def get_u_device(self, dptr: int):
        _lib.check(self.lib.ihm2mpc_get_u_device(self._h, C.c_void_p(dptr)))

    def get_status_device(self, dptr: int):
        _lib.check(self.lib.ihm2mpc_get_status_device(self._h, C.c_void_p(dptr)))

    # ---- plant ----
    def sim_step(self, x, u, model: int = 0, M_sim: int = 100):
        x = _f64(x, (self.B, NX), "x"); u = _f64(u, (self.B, NU), "u")
        out = np.empty((self.B, NX))
        _lib.check(self.lib.ihm2mpc_sim_step(self._h, int(model), int(M_sim), _ptr(x), _ptr(u), _ptr(out)))
        return out

    # ---- Cartesian side of the ROS stack (plants of sim_node.cpp, Track::project) ----
    def set_track_geometry(self, X_ref, Y_ref, phi_ref):
        """Centre line on the ``s_ref`` grid, ``(ntracks, nknots)`` each (the track-file columns of ``tracks.cpp:132-181``)."""
        arrs = [_f64(np.atleast_2d(a), (self.ntracks, self.nknots), n) for a, n in ((X_ref, "X_ref"), (Y_ref, "Y_ref"), (phi_ref, "phi_ref"))]
        _lib.check(self.lib.ihm2mpc_set_track_geometry(self._h, *[_ptr(a) for a in arrs]))

    def sim_step_cart(self, x, u, model: int = -3, M_sim: int = 10, dt_sim: float = 0.01, n_steps: int = 5, v_dyn: float = 3.0):
        """``n_steps`` plant steps of ``dt_sim`` under a constant ``u``; model 3 = kin6, 4 = dyn6, -3 = the node's switch."""
        x = _f64(x, (self.B, NX), "x"); u = _f64(u, (self.B, NU), "u")
        out = np.empty((self.B, NX))
        _lib.check(self.lib.ihm2mpc_sim_step_cart(self._h, int(model), int(M_sim), float(dt_sim), int(n_steps), float(v_dyn), _ptr(x), _ptr(u), _ptr(out)))
        return out

    def project(self, x_cart, s_guess, s_tol: float = 2.0):
        """Cartesian states ``(B,8)`` -> Frenet states ``(B,8)`` and the next projection guess ``(B,)``."""
        xc = _f64(x_cart, (self.B, NX), "x_cart"); sg = _f64(s_guess, (self.B,), "s_guess").copy()
        out = np.empty((self.B, NX))
        _lib.check(self.lib.ihm2mpc_project(self._h, _ptr(xc), _ptr(sg), float(s_tol), _ptr(out)))
        return out, sg

    def set_cart_state(self, x_cart, s_guess):
        _lib.check(self.lib.ihm2mpc_set_cart_state(self._h, _ptr(_f64(x_cart, (self.B, NX), "x_cart")), _ptr(_f64(s_guess, (self.B,), "s_guess"))))

    def get_cart_state(self):
        xc = np.empty((self.B, NX)); sg = np.empty(self.B)
        _lib.check(self.lib.ihm2mpc_get_cart_state(self._h, _ptr(xc), _ptr(sg)))
        return xc, sg

    def sim_advance_cart(self, model: int = -3, M_sim: int = 10, dt_sim: float = 0.01, n_steps: int = 5, v_dyn: float = 3.0, s_tol: float = 2.0):
        """On device: x_cart <- plant(x_cart, u0 of the last solve); x0 <- project(x_cart)."""
        _lib.check(self.lib.ihm2mpc_sim_advance_cart(self._h, int(model), int(M_sim), float(dt_sim), int(n_steps), float(v_dyn), float(s_tol)))

    def sim_advance(self, model: int = 0, M_sim: int = 100):
        """x0 <- plant(x0, u0 of the last solve), on device."""
        _lib.check(self.lib.ihm2mpc_sim_advance(self._h, int(model), int(M_sim)))

    def alloc_pinned(self, shape):
        """float64 array in pinned host memory (for :meth:`get_u0_async`); freed with the solver."""
        n = int(np.prod(shape))
        ptr = C.c_void_p()
        _lib.check(self.lib.ihm2mpc_host_alloc(C.c_uint64(n * 8), C.byref(ptr)))
        self._pinned = getattr(self, "_pinned", []) + [ptr]
        return np.ctypeslib.as_array(C.cast(ptr, _lib.c_double_p), shape=(n,)).reshape(shape)

    def get_u0_async(self, out):
        """Enqueue the copy of ``u0`` (B,2) into the pinned array ``out``; valid after :meth:`synchronize`."""
        if out.shape != (self.B, NU) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 (B, 2) array from alloc_pinned")
        _lib.check(self.lib.ihm2mpc_get_u0_async(self._h, _ptr(out)))

    def set_lap_wrap(self, enable: bool = True):
        """Keep endless closed loops inside the three-lap track tables: cars past ``s = L`` are moved back by one lap."""
        _lib.check(self.lib.ihm2mpc_set_lap_wrap(self._h, int(bool(enable))))

    def set_active(self, active=None):
        """Plant mask for ``sim_advance`` / ``step``: instances with ``active[b] == 0`` keep their ``x0``; ``None`` = all."""
        if active is None:
            _lib.check(self.lib.ihm2mpc_set_active(self._h, None))
        else:
            a = np.ascontiguousarray(active, dtype=np.int32)
            if a.shape != (self.B,):
                raise ValueError(f"active must have shape ({self.B},)")
            _lib.check(self.lib.ihm2mpc_set_active(self._h, a.ctypes.data_as(_lib.c_int32_p)))

    def step(self, s_target: float, model: int = 0, M_sim: int = 100):
        """One MiL iteration on the device: ``sim_advance`` + ``prepare_step`` + one RTI iteration, with the plant step
        overlapped with the linearisation.  Asynchronous; read results with ``get_u0`` / ``get_status``."""
        _lib.check(self.lib.ihm2mpc_step(self._h, int(model), int(M_sim), float(s_target)))

    # ---- single stage of a single instance ----
    def _set_stage(self, i, stage, field, value):
        v = _f64(np.ravel(value))
        _lib.check(self.lib.ihm2mpc_set_stage(self._h, i, stage, field.encode(), _ptr(v), v.size))

    def _get_stage(self, i, stage, field, n):
        out = np.empty(n)
        _lib.check(self.lib.ihm2mpc_get_stage(self._h, i, stage, field.encode(), _ptr(out), n))
        return out


class AcadosOcpSolver:
    """Per-instance shim with the method names and semantics of ``acados_template.AcadosOcpSolver`` as
    the reference uses them.  Built from an OCP it owns a batch of one; ``batch[i]`` gives a view."""

    def __init__(self, ocp: AcadosOcp, json_file: str | None = None, verbose: bool = False, *, s_ref=None, kappa_ref=None,
                 device: int = 0):
        n_half = ocp.dims.np // 2
        if s_ref is None:           # placeholder table until set(i, "p", p) arrives (python/main.py:249-252)
            s_ref = np.arange(n_half, dtype=np.float64)
            kappa_ref = np.zeros(n_half)
        self.batch = BatchedOcpSolver(ocp, 1, s_ref, kappa_ref, device=device)
        self.i = 0
        self._owner = True

    @classmethod
    def _view(cls, batch: BatchedOcpSolver, i: int) -> "AcadosOcpSolver":
        self = cls.__new__(cls)
        self.batch, self.i, self._owner = batch, i, False
        return self

    @property
    def N(self):
        return self.batch.N

    # -- set -------------------------------------------------------------------------------------
    def set(self, stage: int, field: str, value) -> None:
        b = self.batch
        if field == "p":
            p = _f64(np.ravel(value))
            if p.size != 2 * b.nknots:
                raise ValueError(f"p must have {2 * b.nknots} entries [s_ref; kappa_ref], got {p.size}")
            t = int(b._track_id[self.i])       # the table is per track, not per stage: same p for all stages
            if not (np.array_equal(b._s_ref[t], p[:b.nknots]) and np.array_equal(b._kappa_ref[t], p[b.nknots:])):
                b._s_ref[t], b._kappa_ref[t] = p[:b.nknots], p[b.nknots:]
                b._push_tracks()
        elif field in ("lbx", "ubx"):
            self.constraints_set(stage, field, value)
        elif field in ("x", "u", "pi", "lam"):
            b._set_stage(self.i, stage, field, value)
        elif field in ("yref", "y_ref"):
            v = np.ravel(value)
            if stage == b.N:
                # the C++ node hands a 12-vector at the terminal stage too (quirk Q8): keep the first 8
                b._set_stage(self.i, 0, "yref_e", v[:NX])
            else:
                b._set_stage(self.i, stage, "yref", v)
        else:
            raise Exception(f"AcadosOcpSolver.set(): '{field}' is not a valid argument.")

    def cost_set(self, stage: int, field: str, value, api: str = "warn") -> None:
        if field != "W":
            raise Exception(f"AcadosOcpSolver.cost_set(): '{field}' is not a valid argument.")
        b = self.batch
        W = np.asarray(value, dtype=np.float64)
        if stage == b.N:
            if W.shape != (NX, NX):
                raise Exception(f"terminal W must be {NX}x{NX}")
            b.data.W_e = W.copy()
        else:
            if W.shape != (NY, NY):
                raise Exception(f"stage W must be {NY}x{NY}")
            b.data.W[stage] = W
        b._push_weights()

    def constraints_set(self, stage: int, field: str, value, api: str = "warn") -> None:
        b, d, c = self.batch, self.batch.data, self.batch.ocp.constraints
        v = np.asarray(value, dtype=np.float64)
        if field in ("lbx", "ubx") and stage == 0:
            b._set_stage(self.i, 0, field, v)      # initial-state equality (python/main.py:299-300)
            return
        if field in ("lbx", "ubx"):
            idx = np.asarray(c.idxbx_e if stage == b.N else c.idxbx, dtype=int)
            getattr(d, field)[stage, idx] = v
        elif field in ("lbu", "ubu"):
            getattr(d, field)[stage, np.asarray(c.idxbu, dtype=int)] = v
        elif field in ("C", "D"):
            getattr(d, field)[stage, :v.shape[0]] = v
        elif field in ("lg", "ug"):
            getattr(d, field)[stage, :v.size] = v
        else:
            raise Exception(f"AcadosOcpSolver.constraints_set(): '{field}' is not a valid argument.")
        b._push_bounds()

    # -- solve / get -----------------------------------------------------------------------------
    def solve(self) -> int:
        """Runs the solver (for a view: the whole batch) and returns this instance's status."""
        return int(self.batch.solve()[self.i])

    def get(self, stage: int, field: str) -> np.ndarray:
        sizes = {"x": NX, "u": NU, "pi": NX, "lam": NLAM}
        if field not in sizes:
            raise Exception(f"AcadosOcpSolver.get(): '{field}' is not a valid argument.")
        return self.batch._get_stage(self.i, stage, field, sizes[field])

    def get_stats(self, field: str):
        if field == "residuals":
            return self.batch.get_residuals()[self.i]
        if field == "qp_iter":
            return int(self.batch.get_qp_iter()[self.i])
        if field in ("sqp_iter", "alpha"):
            if self.batch.data.nlp_solver_type != "SQP":
                return 1 if field == "sqp_iter" else 1.0
            return self.batch.get_sqp_stats()[field][self.i].item()
        if field == "time_tot":
            return self.batch.get_timings()["total_ms"] * 1e-3
        raise Exception(f"AcadosOcpSolver.get_stats(): '{field}' is not a valid argument.")

    def get_status(self) -> int:
        return int(self.batch.get_status()[self.i])


def get_acados_solver(ocp: AcadosOcp, opts: AcadosOcpOptions, gen_code_dir: str | None = None, **kw) -> AcadosOcpSolver:
    """``python/mpc.py:104-113``; nothing is generated or compiled, ``gen_code_dir`` is accepted and ignored."""
    ocp.solver_options = opts
    return AcadosOcpSolver(ocp, json_file=None, verbose=False, **kw)
