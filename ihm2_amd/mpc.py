"""Host-side mirror of the reference's ``python/mpc.py``: ``ModelBounds``, ``get_acados_ocp``, ``get_acados_solver`` (re-exported from
``ocp.py`` / ``solver.py``) and ``get_ipopt_solver`` (``python/mpc.py:116-216``), the full-NLP multiple-shooting formulation.

``get_ipopt_solver`` builds the SAME nonlinear programme -- decision vector ``[vec(X); vec(U)]`` with ``X`` 8 x (Nf+1), ``U`` 2 x Nf
(column-major), cost ``sum_k x_k'Q x_k + u_k'R u_k + x_N'Qf x_N`` (no reference, no 1/2), state boxes on ``(n, v_x, T, delta)`` at every
stage, input boxes, multiple-shooting equalities ``Phi(x_k, u_k) - x_{k+1} = 0`` and the rate rows ``x[6:8] - u`` -- and returns
``(solver, lbx, ubx, lbg, ubg)`` as the reference does.  ``solver`` is called like a CasADi ``nlpsol`` function,
``solver(x0=w0, lbx=, ubx=, lbg=, ubg=)``, and returns ``{"x", "f", "g", "lam_g", "lam_x"}``.  Where the reference hands the programme to
IPOPT, this one is solved ON THE GPU by the SQP mode of ``libihm2mpc.so`` run to convergence (Gauss-Newton Hessian -- exact for this
least-squares cost --, interior-point QPs, merit line search): a local solution of the same KKT system, the accuracy reference for the
RTI / two-iteration SQP paths.  There is no CPU fallback.

Named deviations from the text of ``python/mpc.py:131-139``:
* the discretisation is classical RK4 with ``rk4_substeps`` sub-steps per interval (default 25, as everywhere in this build): the
  reference's single step of ``dt = 0.05`` is unstable on the throttle lag ``t_T = 1e-3`` (amplification 2.4e5 per step), and
  its ``k3 = f(x + dt k2)`` (quirk Q2: ``dt`` where the tableau has ``dt / 2``) is not reproduced -- SURVEY section 8c;
* the first state must be fixed by the caller (``lbx[:8] == ubx[:8]``, the way the commented call site ``python/main.py:377-388`` uses it):
  the shooting solver has the initial-state equality built in.
"""
from __future__ import annotations

import numpy as np

from .constants import NU, NX, t_T, t_delta
from .ocp import (AcadosOcp, AcadosOcpOptions, ModelBounds, get_acados_model_from_explicit_dynamics, get_acados_ocp)  # noqa: F401
from .solver import AcadosOcpSolver, BatchedOcpSolver, get_acados_solver  # noqa: F401

__all__ = ["ModelBounds", "get_acados_ocp", "get_acados_solver", "get_ipopt_solver", "nlp_ocp", "NlpSolver"]

_BX = np.array([1, 3, 6, 7])      # bounded states: n, v_x, T, delta (python/mpc.py:169-179)


class NlpSolver:
    """The callable ``get_ipopt_solver`` returns: CasADi's ``nlpsol`` calling convention over the GPU SQP solver (batch of one)."""

    def __init__(self, ocp: AcadosOcp, s_ref, kappa_ref, Q, R, Qf, device: int = 0):
        self.Nf = ocp.dims.N
        self.Q, self.R, self.Qf = (np.asarray(a, dtype=np.float64) for a in (Q, R, Qf))
        self._batch = BatchedOcpSolver(ocp, 1, s_ref, kappa_ref, device=device)
        self._view = self._batch[0]
        self._stats = {"success": False, "return_status": "not run", "iter_count": 0}
        self.n_w, self.n_g = NX * (self.Nf + 1) + NU * self.Nf, (NX + NU) * self.Nf

    # ---- (un)packing of CasADi's column-major vectors ----
    def _split(self, w):
        w = np.asarray(w, dtype=np.float64).reshape(-1)
        if w.size != self.n_w:
            raise ValueError(f"the decision vector has {self.n_w} entries (8 x {self.Nf + 1} states, then 2 x {self.Nf} inputs), got {w.size}")
        ns = NX * (self.Nf + 1)
        return w[:ns].reshape(self.Nf + 1, NX).copy(), w[ns:].reshape(self.Nf, NU).copy()

    def __call__(self, x0=None, lbx=None, ubx=None, lbg=None, ubg=None, p=None, lam_x0=None, lam_g0=None):
        Nf, b, v = self.Nf, self._batch, self._view
        if lbx is None or ubx is None or lbg is None or ubg is None:
            raise ValueError("lbx, ubx, lbg, ubg are required (the vectors get_ipopt_solver returned, with the initial state fixed)")
        X, U = self._split(np.zeros(self.n_w) if x0 is None else x0)
        Xl, Ul = self._split(lbx); Xu, Uu = self._split(ubx)
        if not np.array_equal(Xl[0], Xu[0]) or not np.all(np.isfinite(Xl[0])):
            raise ValueError("the initial state must be fixed: lbx[:8] == ubx[:8] (python/main.py:377-388 sets it from the measured state)")
        free = np.setdiff1d(np.arange(NX), _BX)
        if np.any(np.isfinite(Xl[1:, free])) or np.any(np.isfinite(Xu[1:, free])):
            raise NotImplementedError("state bounds other than on (n, v_x, T, delta) are not part of this programme (python/mpc.py:166-179)")
        lbg = np.asarray(lbg, dtype=np.float64).reshape(-1); ubg = np.asarray(ubg, dtype=np.float64).reshape(-1)
        if lbg.size != self.n_g or ubg.size != self.n_g:
            raise ValueError(f"lbg / ubg have {self.n_g} entries (8 x {Nf} shooting equalities, then 2 x {Nf} rate rows)")
        if np.any(lbg[:NX * Nf] != 0.0) or np.any(ubg[:NX * Nf] != 0.0):
            raise NotImplementedError("the shooting constraints are equalities: lbg = ubg = 0 on the first 8 Nf rows")
        llin, ulin = lbg[NX * Nf:].reshape(Nf, NU), ubg[NX * Nf:].reshape(Nf, NU)
        # bounds, stage by stage (what AcadosOcpSolver.constraints_set(stage, ...) writes), one upload
        d = b.data
        d.lbx[1:, :] = -np.inf; d.ubx[1:, :] = np.inf
        d.lbx[1:, _BX], d.ubx[1:, _BX] = Xl[1:, _BX], Xu[1:, _BX]
        d.lbu[:], d.ubu[:] = Ul, Uu
        # this build's rows are u - x[6:8] (python/mpc.py:92-99); the programme's are x[6:8] - u (python/mpc.py:159-161)
        d.lg[:, :NU], d.ug[:, :NU] = -ulin, -llin
        b._push_bounds()
        X[0] = Xl[0]
        b.set_x0(Xl[:1]); b.set_x(X[None]); b.set_u(U[None])
        b.set_yref(np.zeros((1, Nf, NX + 2 * NU))); b.set_yref_e(np.zeros((1, NX)))
        b.set_multipliers(None, None)
        status = int(b.solve()[0])
        st = b.get_sqp_stats()
        Xs, Us = b.get_x()[0], b.get_u()[0]
        pi, lam = b.get_multipliers()
        b.linearize()
        A, _, defect = b.get_linearization()                  # d Phi / dx and Phi(x_k, u_k) - x_{k+1} at the returned point
        f = float(np.einsum("ki,ij,kj->", Xs[:-1], self.Q, Xs[:-1]) + np.einsum("ki,ij,kj->", Us, self.R, Us) + Xs[-1] @ self.Qf @ Xs[-1])
        self._stats = {"success": status == 0, "return_status": {0: "Solve_Succeeded", 2: "Maximum_Iterations_Exceeded"}.get(status, f"Error_{status}"),
                       "iter_count": int(st["sqp_iter"][0]), "status": status, "residuals": b.get_residuals()[0].tolist()}
        # Multipliers in CasADi's convention, L = f + lam_g'g + lam_x'w (positive at an active upper bound).  The solver's Lagrangian is
        # f + sum pi_{k+1}'(Phi_k - x_{k+1}) - sum (lam_lower - lam_upper)'(row): the costates ARE lam_g of the shooting rows; the rate
        # rows change sign with the row (u - x_act here, x_act - u there).
        lam = lam[0]
        lam_lin = lam[:Nf, 10:12] - lam[:Nf, 24:26]
        lam_X = lam[:, 14:22] - lam[:, 0:8]
        lam_U = lam[:Nf, 22:24] - lam[:Nf, 8:10]
        # the fixed initial state: its multiplier is what makes the Lagrangian stationary in x_0
        gx0 = 2.0 * self.Q @ Xs[0] + A[0, 0].T @ pi[0, 1]
        gx0[6:8] += lam_lin[0]
        lam_X[0] = -gx0
        return {"x": np.concatenate((Xs.reshape(-1), Us.reshape(-1))), "f": f,
                "g": np.concatenate((defect[0].reshape(-1), (Xs[:-1, 6:8] - Us).reshape(-1))),
                "lam_g": np.concatenate((pi[0, 1:].reshape(-1), lam_lin.reshape(-1))),
                "lam_x": np.concatenate((lam_X.reshape(-1), lam_U.reshape(-1)))}

    def stats(self) -> dict:
        return dict(self._stats)

    def free(self):
        self._batch.free()


def nlp_ocp(continuous_model_fn, Nf: int, model_bounds: ModelBounds, dt: float, n_params: int, Q, R, Qf,
            rk4_substeps: int = 25, max_iter: int = 100, tol: float = 1e-6) -> AcadosOcp:
    """The programme of ``python/mpc.py:116-216`` as an ``AcadosOcp`` for the SQP mode (what ``get_ipopt_solver`` solves)."""
    mb = model_bounds
    Q, R, Qf = (np.asarray(a, dtype=np.float64) for a in (Q, R, Qf))
    if Q.shape != (NX, NX) or R.shape != (NU, NU) or Qf.shape != (NX, NX):
        raise ValueError("Q, Qf are 8 x 8 and R is 2 x 2")
    model = get_acados_model_from_explicit_dynamics("ihm2_nlp", continuous_model_fn, NX, NU, n_params)
    ocp = get_acados_ocp(model, Nf, mb.n_max, mb.v_x_max, mb.T_max, mb.delta_max, mb.T_dot_max, mb.delta_dot_max)
    c = ocp.constraints
    c.lbx = np.array([-mb.n_max, mb.v_x_min, -mb.T_max, -mb.delta_max])
    # the programme bounds the SAME four states at the terminal stage (python/mpc.py:166-179; the acados OCP's terminal index set is quirk Q1)
    c.idxbx_e, c.lbx_e, c.ubx_e = c.idxbx.copy(), c.lbx.copy(), c.ubx.copy()
    # cost sum x'Qx + u'Ru (no 1/2, no time-step factor) as the least-squares cost 1/2 |y|^2_W over y = [x; u; x_act - u]
    W = np.zeros((NX + 2 * NU, NX + 2 * NU)); W[:NX, :NX] = 2.0 * Q; W[NX:NX + NU, NX:NX + NU] = 2.0 * R
    ocp.cost.W, ocp.cost.W_e = W, 2.0 * Qf
    o = AcadosOcpOptions()
    o.tf = Nf * dt
    o.nlp_solver_type, o.nlp_solver_max_iter, o.globalization = "SQP", int(max_iter), "MERIT_BACKTRACKING"
    o.integrator_type, o.sim_method_num_steps = "ERK", int(rk4_substeps)
    o.cost_scale_stage = 1.0
    o.nlp_tol = float(tol)
    # the QPs' complementarity floor bounds what the outer iteration can reach: three digits below the NLP tolerance
    o.qp_tol = max(1e-3 * float(tol), 1e-11)
    o.qp_solver_iter_max = 50
    ocp.solver_options = o
    return ocp


def get_ipopt_solver(continuous_model_fn, Nf: int, model_bounds: ModelBounds, dt: float, s_ref, kappa_ref, Q, R, Qf,
                     rk4_substeps: int = 25, max_iter: int = 100, tol: float = 1e-6, device: int = 0):
    """``python/mpc.py:116-216``: returns ``(solver, lbx, ubx, lbg, ubg)``; see the module docstring for what ``solver`` is here."""
    mb = model_bounds
    ocp = nlp_ocp(continuous_model_fn, Nf, mb, dt, 2 * np.asarray(s_ref).reshape(-1).size, Q, R, Qf, rk4_substeps, max_iter, tol)
    solver = NlpSolver(ocp, s_ref, kappa_ref, Q, R, Qf, device=device)

    # the bound vectors of python/mpc.py:166-198, column-major like the decision vector
    lstate = np.full((NX, Nf + 1), -np.inf); ustate = np.full((NX, Nf + 1), np.inf)
    lstate[1], lstate[3], lstate[6], lstate[7] = -mb.n_max, mb.v_x_min, -mb.T_max, -mb.delta_max
    ustate[1], ustate[3], ustate[6], ustate[7] = mb.n_max, mb.v_x_max, mb.T_max, mb.delta_max
    lcontrol = np.empty((NU, Nf)); ucontrol = np.empty((NU, Nf))
    lcontrol[0], lcontrol[1] = -mb.T_max, -mb.delta_max
    ucontrol[0], ucontrol[1] = mb.T_max, mb.delta_max
    lbx = np.concatenate((np.ravel(lstate, "F"), np.ravel(lcontrol, "F")))
    ubx = np.concatenate((np.ravel(ustate, "F"), np.ravel(ucontrol, "F")))
    ulin = np.empty((NU, Nf)); ulin[0], ulin[1] = t_T * mb.T_dot_max, t_delta * mb.delta_dot_max
    lbg = np.concatenate((np.zeros(NX * Nf), np.ravel(-ulin, "F")))
    ubg = np.concatenate((np.zeros(NX * Nf), np.ravel(ulin, "F")))
    return solver, lbx, ubx, lbg, ubg
