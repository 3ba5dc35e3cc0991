"""CPU ORACLE binding (test infrastructure, NOT the product).

ctypes wrapper over ``oracle/libihm2_oracle.so`` (built by ``make -C oracle``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libihm2_oracle.so")
# the same sources built with fifteen constraint rows per stage: row 14 = the lateral-acceleration row (ihm2_oracle.h, ORC_NC)
_LIB15_PATH = os.path.join(_HERE, "libihm2_oracle_nc15.so")
NX, NU, NZ, NY, NC, NG, NH = 8, 2, 10, 12, 14, 2, 2
MODEL_FKIN6, MODEL_FDYN6, MODEL_FDYN6U, MODEL_KIN6, MODEL_DYN6 = 0, 1, 2, 3, 4
INTEG_RK4 = 0
INTEG_IRK_GL4 = 1      # acados IRK, GAUSS_LEGENDRE, 4 stages (python/main.py:234-236)
INTEG_IRK_RADAU4 = 2   # acados IRK, GAUSS_RADAU_IIA, 4 stages (python/main.py:395-400, python/sim.py:28-33)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class _Problem(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("M", C.c_int), ("model", C.c_int), ("integrator", C.c_int),
        ("dt", C.c_double), ("cost_scale_stage", C.c_double),
        ("ntracks", C.c_int), ("nknots", C.c_int), ("s_ref", _dp), ("kappa_ref", _dp),
        ("W", _dp), ("W_e", _dp), ("lbx", _dp), ("ubx", _dp), ("lbu", _dp), ("ubu", _dp),
        ("C", _dp), ("D", _dp), ("lg", _dp), ("ug", _dp), ("soft_z", _dp), ("soft_Z", _dp),
        ("path_on", C.c_int), ("car_L", C.c_double), ("car_W", C.c_double), ("widths", _dp),
        ("lh", C.c_double * NH), ("uh", C.c_double * NH),
        ("ipm_iter_max", C.c_int), ("ipm_tol", C.c_double), ("ipm_mu0", C.c_double), ("ipm_tau0", C.c_double),
        ("alat_on", C.c_int), ("alat_lb", C.c_double), ("alat_ub", C.c_double),
    ]


class _SqpOpts(C.Structure):
    """``orc_sqp_opts`` (ihm2_oracle.h)."""
    _fields_ = [("max_iter", C.c_int), ("globalization", C.c_int), ("use_sufficient_descent", C.c_int), ("full_step_dual", C.c_int),
                ("tol", C.c_double * 4), ("alpha_min", C.c_double), ("alpha_reduction", C.c_double), ("eps_sufficient_descent", C.c_double)]


def build(force: bool = False) -> str:
    # make decides by the time stamps: a library older than its sources is rebuilt (a stale one once hid a change of the iteration from its own test)
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    except (OSError, subprocess.CalledProcessError):
        if force or not (os.path.exists(_LIB_PATH) and os.path.exists(_LIB15_PATH)):
            raise
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_kappa.restype = C.c_double
        _lib.orc_qp_solve.restype = C.c_int
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_alat.restype = C.c_double
    return _lib


_lib15 = None


def lib15():
    """The fifteen-row build (row 14 = lateral acceleration): arrays over the constraint rows are 15 / 30 wide there."""
    global _lib15
    if _lib15 is None:
        build()
        _lib15 = C.CDLL(_LIB15_PATH)
        _lib15.orc_qp_solve.restype = C.c_int
    return _lib15


def alat(x):
    """``a_lat`` of the kinematic model at x (8) and its gradient (8) -- ``orc_alat``."""
    x, xp = _d(x)
    g = np.zeros(NX)
    return float(lib().orc_alat(xp, g.ctypes.data_as(_dp))), g


def widen_rows(a28, a2):
    """(.., 28) arrays over the fourteen rows + (.., 2) of the lateral-acceleration row -> the (.., 30) layout of the fifteen-row build."""
    a28, a2 = np.asarray(a28), np.asarray(a2)
    return np.concatenate([a28[..., :NC], a2[..., 0:1], a28[..., NC:], a2[..., 1:2]], axis=-1)


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def f(model, x, u, s_ref, kappa_ref):
    x, xp = _d(x); u, up = _d(u); s, sp = _d(s_ref); k, kp = _d(kappa_ref)
    out = np.zeros(NX)
    lib().orc_f(C.c_int(model), xp, up, sp, kp, C.c_int(len(s)), out.ctypes.data_as(_dp))
    return out


def jac(model, x, u, s_ref, kappa_ref, complex_step=False):
    x, xp = _d(x); u, up = _d(u); s, sp = _d(s_ref); k, kp = _d(kappa_ref)
    out = np.zeros(NX); J = np.zeros((NX, NZ))
    fn = lib().orc_jac_cs if complex_step else lib().orc_jac
    fn(C.c_int(model), xp, up, sp, kp, C.c_int(len(s)), out.ctypes.data_as(_dp), J.ctypes.data_as(_dp))
    return out, J


def kappa(s_ref, kappa_ref, s):
    sr, sp = _d(s_ref); k, kp = _d(kappa_ref)
    dk = C.c_double(0.0)
    v = lib().orc_kappa(sp, kp, C.c_int(len(sr)), C.c_double(s), C.byref(dk))
    return v, dk.value


def rk4_sens(model, x, u, s_ref, kappa_ref, dt, M, integrator=INTEG_RK4):
    x, xp = _d(x); u, up = _d(u); s, sp = _d(s_ref); k, kp = _d(kappa_ref)
    xn = np.zeros(NX); A = np.zeros((NX, NX)); Bm = np.zeros((NX, NU))
    lib().orc_rk4_sens(C.c_int(model), C.c_int(integrator), xp, up, sp, kp, C.c_int(len(s)), C.c_double(dt), C.c_int(M),
                       xn.ctypes.data_as(_dp), A.ctypes.data_as(_dp), Bm.ctypes.data_as(_dp))
    return xn, A, Bm


def rk4(model, x, u, s_ref, kappa_ref, dt, M, integrator=INTEG_RK4):
    x, xp = _d(x); u, up = _d(u); s, sp = _d(s_ref); k, kp = _d(kappa_ref)
    xn = np.zeros(NX)
    lib().orc_rk4(C.c_int(model), C.c_int(integrator), xp, up, sp, kp, C.c_int(len(s)), C.c_double(dt), C.c_int(M),
                  xn.ctypes.data_as(_dp))
    return xn


def qp_solve(H, g, A, Bm, b, dx0, R, dl, du, iter_max=50, tol=1e-6, mu0=0.03, tau0=0.1, soft_z=None, soft_Z=None):
    N = A.shape[0]
    H, Hp = _d(H); g, gp = _d(g); A, Ap = _d(A); Bm, Bp = _d(Bm); b, bp = _d(b); dx0, xp = _d(dx0)
    R, Rp = _d(R); dl, lp = _d(dl); du, up = _d(du)
    dz = np.zeros((N + 1, NZ)); pi = np.zeros((N + 1, NX)); lam = np.zeros((N + 1, 2 * NC)); t = np.zeros((N + 1, 2 * NC))
    sl = np.zeros((N + 1, 2 * NC))
    stats = np.zeros(8); iters = C.c_int(0)
    if soft_Z is None:
        szp = sZp = None
    else:
        sz, szp = _d(soft_z); sZ, sZp = _d(soft_Z)
        assert sz.shape == (N + 1, 2 * NC) and sZ.shape == (N + 1, 2 * NC)
    fn = lib().orc_qp_solve_soft
    fn.restype = C.c_int
    st = fn(C.c_int(N), Hp, gp, Ap, Bp, bp, xp, Rp, lp, up, szp, sZp, C.c_int(iter_max), C.c_double(tol),
            C.c_double(mu0), C.c_double(tau0), dz.ctypes.data_as(_dp), pi.ctypes.data_as(_dp),
            lam.ctypes.data_as(_dp), t.ctypes.data_as(_dp), sl.ctypes.data_as(_dp), stats.ctypes.data_as(_dp), C.byref(iters))
    return dict(status=st, dz=dz, pi=pi, lam=lam, t=t, sl=sl, stats=stats, iters=iters.value)


def project(s_ref, X_ref, Y_ref, phi_ref, X, Y, s_guess, s_tol=2.0):
    sr, sp = _d(s_ref); Xr, Xp = _d(X_ref); Yr, Yp = _d(Y_ref); pr, pp = _d(phi_ref)
    out = [C.c_double(0.0) for _ in range(4)]
    lib().orc_project(sp, Xp, Yp, pp, C.c_int(len(sr)), C.c_double(X), C.c_double(Y), C.c_double(s_guess), C.c_double(s_tol),
                      *[C.byref(o) for o in out])
    return tuple(o.value for o in out)


def cart_to_frenet(s_ref, X_ref, Y_ref, phi_ref, x_cart, s_guess, s_tol=2.0, track_id=None):
    """Tables (ntracks, nk); x_cart (B,8); s_guess (B) -> (x_frenet (B,8), next s_guess (B))."""
    sr, sp = _d(np.atleast_2d(s_ref)); Xr, Xp = _d(np.atleast_2d(X_ref)); Yr, Yp = _d(np.atleast_2d(Y_ref)); pr, pp = _d(np.atleast_2d(phi_ref))
    xc, xcp = _d(x_cart)
    B = xc.shape[0]
    sg = np.array(s_guess, dtype=np.float64).copy()
    tid, tp = _i(np.zeros(B, dtype=np.int32) if track_id is None else track_id)
    xf = np.zeros((B, NX))
    lib().orc_cart_to_frenet(C.c_int(B), C.c_int(sr.shape[0]), C.c_int(sr.shape[1]), sp, Xp, Yp, pp, tp, xcp, sg.ctypes.data_as(_dp),
                             C.c_double(s_tol), xf.ctypes.data_as(_dp))
    return xf, sg


def sim_step_cart(x, u, model, M, dt=0.05, v_dyn=3.0):
    """Cartesian plant: model MODEL_KIN6, MODEL_DYN6 or -3 (speed switch + no reversing, src/ihm2/src/sim_node.cpp:197-257)."""
    x, xp = _d(x); u, up = _d(u)
    xn = np.zeros_like(x)
    lib().orc_sim_step_cart(C.c_int(x.shape[0]), C.c_int(model), C.c_int(M), C.c_double(dt), C.c_double(v_dyn), xp, up, xn.ctypes.data_as(_dp))
    return xn


def f_dyn10(x, u, s_ref, kappa_ref):
    """Explicit ``fdyn10`` derivative (python/models.py:609-801 solved for xdot); x (B,15), u (B,5)."""
    x, _ = _d(np.atleast_2d(x)); u, _ = _d(np.atleast_2d(u)); sr, srp = _d(s_ref); kr, krp = _d(kappa_ref)
    out = np.zeros_like(x)
    for b in range(x.shape[0]):
        lib().orc_f_dyn10(x[b].ctypes.data_as(_dp), u[b].ctypes.data_as(_dp), srp, krp, C.c_int(sr.size), out[b].ctypes.data_as(_dp))
    return out


def sim_step_dyn10(x, u, s_ref, kappa_ref, M, dt=0.05):
    """``fdyn10`` plant step, RK4 x M over dt; x (B,15), u (B,5)."""
    x, xp = _d(np.atleast_2d(x)); u, up = _d(np.atleast_2d(u)); sr, srp = _d(s_ref); kr, krp = _d(kappa_ref)
    xn = np.zeros_like(x)
    lib().orc_sim_step_dyn10(C.c_int(x.shape[0]), C.c_int(M), C.c_double(dt), xp, up, srp, krp, C.c_int(sr.size), xn.ctypes.data_as(_dp))
    return xn


def jac_dyn10(x, u, s_ref, kappa_ref):
    """``fdyn10``: (f (15), d f / d x (15, 15)) at one point, by dual-number evaluation of the model's formulas."""
    x, xp = _d(x); u, up = _d(u); sr, srp = _d(s_ref); kr, krp = _d(kappa_ref)
    f = np.zeros(15); J = np.zeros((15, 15))
    lib().orc_jac_dyn10(xp, up, srp, krp, C.c_int(sr.size), f.ctypes.data_as(_dp), J.ctypes.data_as(_dp))
    return f, J


def sim_step_dyn10_irk(x, u, s_ref, kappa_ref, M=100, dt=0.05, integrator=INTEG_IRK_RADAU4, newton_iter=10):
    """``fdyn10`` plant step with the reference's integrator (python/main.py:395-400): 4-stage Radau IIA collocation on a grid of at most
    dt / M, every step solved to convergence (at most ``newton_iter`` Newton iterations, else the step is cut); x (B,15), u (B,5)."""
    x, xp = _d(np.atleast_2d(x)); u, up = _d(np.atleast_2d(u)); sr, srp = _d(s_ref); kr, krp = _d(kappa_ref)
    xn = np.zeros_like(x)
    lib().orc_sim_step_dyn10_irk(C.c_int(x.shape[0]), C.c_int(integrator), C.c_int(M), C.c_int(newton_iter), C.c_double(dt), xp, up, srp, krp,
                                 C.c_int(sr.size), xn.ctypes.data_as(_dp))
    return xn


class OracleProblem:
    """Holds the arrays of an ``orc_problem`` alive.  ``desc`` is a plain dict of numpy arrays and
    scalars (the product's ``OcpData.as_dict()`` produces exactly this)."""

    def __init__(self, desc: dict):
        self.N = int(desc["N"]); self.M = int(desc["M"])
        self._keep = {}
        # the lateral-acceleration row: the fifteen-row build of the same sources; lam / sl arrays are then (.., 30) (widen_rows)
        self.alat_on = int(desc.get("alat_on", 0))
        self.nc = NC + 1 if self.alat_on else NC
        self._lib = lib15() if self.alat_on else lib()
        p = _Problem()
        p.N = self.N; p.M = self.M; p.model = int(desc.get("model", 0)); p.integrator = int(desc.get("integrator", 0))
        p.dt = float(desc["dt"]); p.cost_scale_stage = float(desc["cost_scale_stage"])
        s_ref = np.atleast_2d(np.asarray(desc["s_ref"], dtype=np.float64))
        k_ref = np.atleast_2d(np.asarray(desc["kappa_ref"], dtype=np.float64))
        p.ntracks, p.nknots = s_ref.shape
        for name, arr in (("s_ref", s_ref), ("kappa_ref", k_ref), ("W", desc["W"]), ("W_e", desc["W_e"]),
                          ("lbx", desc["lbx"]), ("ubx", desc["ubx"]), ("lbu", desc["lbu"]), ("ubu", desc["ubu"]),
                          ("C", desc["C"]), ("D", desc["D"]), ("lg", desc["lg"]), ("ug", desc["ug"])):
            a, ptr = _d(arr)
            self._keep[name] = a
            setattr(p, name, ptr)
        soft = {k: desc.get(k) for k in ("soft_z", "soft_Z")}
        if self.alat_on:
            p.alat_on = 1; p.alat_lb, p.alat_ub = float(desc["alat_lb"]), float(desc["alat_ub"])
            if soft["soft_Z"] is not None or desc.get("alat_soft_Z") is not None:
                N1 = self.N + 1
                z28 = np.zeros((N1, 2 * NC)) if soft["soft_z"] is None else np.asarray(soft["soft_z"], dtype=np.float64)
                Z28 = np.full((N1, 2 * NC), -1.0) if soft["soft_Z"] is None else np.asarray(soft["soft_Z"], dtype=np.float64)
                az = np.zeros((N1, 2)); aZ = np.full((N1, 2), -1.0)
                if desc.get("alat_soft_Z") is not None:
                    az[1:self.N] = np.asarray(desc["alat_soft_z"], dtype=np.float64); aZ[1:self.N] = np.asarray(desc["alat_soft_Z"], dtype=np.float64)
                soft = {"soft_z": widen_rows(z28, az), "soft_Z": widen_rows(Z28, aZ)}
        if soft["soft_Z"] is not None:
            for name in ("soft_z", "soft_Z"):
                a, ptr = _d(soft[name])
                assert a.shape == (self.N + 1, 2 * self.nc)
                self._keep[name] = a
                setattr(p, name, ptr)
        p.path_on = int(desc.get("path_on", 0))
        if p.path_on:
            p.car_L, p.car_W = float(desc["car_L"]), float(desc["car_W"])
            w, ptr = _d(np.atleast_2d(desc["widths"]))
            assert w.shape == (p.ntracks, 2)
            self._keep["widths"] = w
            p.widths = ptr
            for i in range(NH):
                p.lh[i] = float(desc["lh"][i]); p.uh[i] = float(desc["uh"][i])
        N = self.N
        assert self._keep["W"].shape == (N, NY, NY) and self._keep["lbx"].shape == (N + 1, NX)
        assert self._keep["C"].shape == (N, NG, NX) and self._keep["D"].shape == (N, NG, NU)
        p.ipm_iter_max = int(desc["ipm_iter_max"]); p.ipm_tol = float(desc["ipm_tol"])
        p.ipm_mu0 = float(desc["ipm_mu0"]); p.ipm_tau0 = float(desc["ipm_tau0"])
        self.p = p

    def rti_step(self, x, u, x0, yref, yref_e, track_id=None, pi=None, lam=None, nthreads=0):
        """One RTI iteration; ``x`` (B,N+1,8) and ``u`` (B,N,2) are updated IN PLACE."""
        N = self.N
        assert x.dtype == np.float64 and x.flags.c_contiguous and u.dtype == np.float64 and u.flags.c_contiguous
        B = x.shape[0]
        x0, x0p = _d(x0); yref, yp = _d(yref); yref_e, yep = _d(yref_e)
        tid, tp = _i(np.zeros(B, dtype=np.int32) if track_id is None else track_id)
        if pi is None: pi = np.zeros((B, N + 1, NX))
        if lam is None: lam = np.zeros((B, N + 1, 2 * self.nc))
        assert pi.flags.c_contiguous and lam.flags.c_contiguous and lam.shape[-1] == 2 * self.nc
        status = np.zeros(B, dtype=np.int32); res = np.zeros((B, 4)); qp_iter = np.zeros(B, dtype=np.int32)
        self._lib.orc_rti_step(C.byref(self.p), C.c_int(B), x.ctypes.data_as(_dp), u.ctypes.data_as(_dp), x0p, yp, yep, tp,
                           pi.ctypes.data_as(_dp), lam.ctypes.data_as(_dp), status.ctypes.data_as(_ip),
                           res.ctypes.data_as(_dp), qp_iter.ctypes.data_as(_ip), C.c_int(nthreads))
        return dict(status=status, res=res, qp_iter=qp_iter, pi=pi, lam=lam)

    def sqp_solve(self, x, u, x0, yref, yref_e, track_id=None, pi=None, lam=None, sl=None, max_iter=2, globalization="MERIT_BACKTRACKING",
                  tol=1e-6, alpha_min=0.05, alpha_reduction=0.7, eps_sufficient_descent=1e-4, use_sufficient_descent=False,
                  full_step_dual=False, nthreads=0):
        """Globalised SQP (``orc_sqp_solve``); ``x``, ``u`` (and ``pi``, ``lam``, ``sl`` when given) are updated IN PLACE."""
        N = self.N
        assert x.dtype == np.float64 and x.flags.c_contiguous and u.dtype == np.float64 and u.flags.c_contiguous
        B = x.shape[0]
        x0, x0p = _d(x0); yref, yp = _d(yref); yref_e, yep = _d(yref_e)
        tid, tp = _i(np.zeros(B, dtype=np.int32) if track_id is None else track_id)
        if pi is None: pi = np.zeros((B, N + 1, NX))
        if lam is None: lam = np.zeros((B, N + 1, 2 * NC))
        if sl is None: sl = np.zeros((B, N + 1, 2 * NC))
        assert pi.flags.c_contiguous and lam.flags.c_contiguous and sl.flags.c_contiguous
        o = _SqpOpts()
        o.max_iter = int(max_iter); o.globalization = {"FIXED_STEP": 0, "MERIT_BACKTRACKING": 1}[globalization]
        o.use_sufficient_descent = int(use_sufficient_descent); o.full_step_dual = int(full_step_dual)
        for i, t in enumerate(np.broadcast_to(np.asarray(tol, dtype=np.float64), (4,))): o.tol[i] = float(t)
        o.alpha_min, o.alpha_reduction, o.eps_sufficient_descent = float(alpha_min), float(alpha_reduction), float(eps_sufficient_descent)
        status = np.zeros(B, dtype=np.int32); res = np.zeros((B, 4)); qp_iter = np.zeros(B, dtype=np.int32)
        sqp_iter = np.zeros(B, dtype=np.int32); alpha = np.zeros(B)
        lib().orc_sqp_solve(C.byref(self.p), C.byref(o), C.c_int(B), x.ctypes.data_as(_dp), u.ctypes.data_as(_dp), x0p, yp, yep, tp,
                            pi.ctypes.data_as(_dp), lam.ctypes.data_as(_dp), sl.ctypes.data_as(_dp), status.ctypes.data_as(_ip),
                            res.ctypes.data_as(_dp), qp_iter.ctypes.data_as(_ip), sqp_iter.ctypes.data_as(_ip),
                            alpha.ctypes.data_as(_dp), C.c_int(nthreads))
        return dict(status=status, res=res, qp_iter=qp_iter, sqp_iter=sqp_iter, alpha=alpha, pi=pi, lam=lam, sl=sl)

    def linearize(self, x, u, track_id=None, nthreads=0):
        N = self.N; B = x.shape[0]
        x, xp = _d(x); u, up = _d(u)
        tid, tp = _i(np.zeros(B, dtype=np.int32) if track_id is None else track_id)
        A = np.zeros((B, N, NX, NX)); Bm = np.zeros((B, N, NX, NU)); b = np.zeros((B, N, NX))
        lib().orc_linearize(C.byref(self.p), C.c_int(B), xp, up, tp, A.ctypes.data_as(_dp), Bm.ctypes.data_as(_dp),
                            b.ctypes.data_as(_dp), C.c_int(nthreads))
        return A, Bm, b

    def build_qp(self, x, u, x0, yref, yref_e, track_id=0):
        N = self.N
        x, xp = _d(x); u, up = _d(u); x0, x0p = _d(x0); yref, yp = _d(yref); yref_e, yep = _d(yref_e)
        H = np.zeros((N + 1, NZ, NZ)); g = np.zeros((N + 1, NZ)); A = np.zeros((N, NX, NX)); Bm = np.zeros((N, NX, NU))
        nc = self.nc
        b = np.zeros((N, NX)); dx0 = np.zeros(NX); R = np.zeros((N + 1, nc, NZ)); dl = np.zeros((N + 1, nc)); du = np.zeros((N + 1, nc))
        self._lib.orc_build_qp(C.byref(self.p), xp, up, x0p, yp, yep, C.c_int(track_id), *[a.ctypes.data_as(_dp) for a in (H, g, A, Bm, b, dx0, R, dl, du)])
        return dict(H=H, g=g, A=A, Bm=Bm, b=b, dx0=dx0, R=R, dl=dl, du=du)

    def sim_step(self, x, u, model, M, track_id=None, nthreads=0, integrator=INTEG_RK4):
        B = x.shape[0]
        x, xp = _d(x); u, up = _d(u)
        tid, tp = _i(np.zeros(B, dtype=np.int32) if track_id is None else track_id)
        xn = np.zeros((B, NX))
        lib().orc_sim_step_integ(C.byref(self.p), C.c_int(B), C.c_int(model), C.c_int(integrator), C.c_int(M), xp, up, tp, xn.ctypes.data_as(_dp),
                                 C.c_int(nthreads))
        return xn


def stanley_guess(P, s_ref, kappa_ref, x0, N, M=25, l_R=0.7853):
    """Initial prediction for a batch that starts anywhere on the track: rollout of the model from x0 under Stanley-type feedback
    (StanleyController.compute_control, python/main.py:139-163; torque: P-term on the speed only), inputs clipped to the box and the
    steering-rate row -- the CPU counterpart of ihm2mpc_init_guess (kernels_misc.hip::k_init_guess) for the reference's bounds."""
    B = x0.shape[0]
    x = np.zeros((B, N + 1, 8)); u = np.zeros((B, N, 2)); x[:, 0] = x0
    for k in range(N):
        xk = x[:, k]
        kap = np.interp(xk[:, 0], s_ref, kappa_ref)
        uT = np.clip(90.0 * (x0[:, 3] - xk[:, 3]), -500, 500)
        ud = np.arctan(2 * np.tan(np.arcsin(np.clip(kap * l_R, -0.9, 0.9)))) - 1.8 * xk[:, 2] - np.arctan(5.5 * xk[:, 1] / (2 + xk[:, 3]))
        ud = np.clip(np.clip(ud, xk[:, 7] - 0.02, xk[:, 7] + 0.02), -0.5, 0.5)
        u[:, k] = np.stack([uT, ud], 1)
        x[:, k + 1] = P.sim_step(xk, u[:, k], 0, M)
    return x, u


def prepare_step(N, x0, s_target, x, u):
    """Shift + reference ramp in place on x (B,N+1,8), u (B,N,2); returns yref (B,N,12), yref_e (B,8)."""
    B = x.shape[0]
    assert x.flags.c_contiguous and u.flags.c_contiguous
    x0, x0p = _d(x0)
    yref = np.zeros((B, N, NY)); yref_e = np.zeros((B, NX))
    lib().orc_prepare_step(C.c_int(N), C.c_int(B), x0p, C.c_double(s_target), x.ctypes.data_as(_dp), u.ctypes.data_as(_dp),
                           yref.ctypes.data_as(_dp), yref_e.ctypes.data_as(_dp))
    return yref, yref_e


def num_threads() -> int:
    return int(lib().orc_num_threads())
