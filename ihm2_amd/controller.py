"""Controllers of the ihm2 MiL loop, batched.

* :class:`IHM2Controller` -- the reference's NMPC controller (``python/main.py:166-334``; its refactor
  skeleton ``new_python/controller.py:240-405``): same constructor arguments and defaults, same
  ``compute_control`` semantics (initial-state equality, reference ramp ``s0 + s_target*j/Nf``,
  warm-start shift, solve, accept status in {0, 2}), for ``batch_size`` independent cars at once.  All
  per-step work runs on the GPU (``ihm2mpc_prepare_step`` + ``ihm2mpc_solve``); the reference's ~207
  ctypes calls per step become 3 for any batch size.
* :class:`StanleyController` -- ``python/main.py:99-163`` (P + I torque control, feed-forward +
  heading + lateral-error steering), vectorised over the batch.
* :class:`Controller` -- the abstract base and its ``Config`` as in ``new_python/controller.py:134-159``.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from collections import deque
from dataclasses import dataclass

import numpy as np

from .constants import NU, NX, l_R
from .ocp import (AcadosOcpOptions, ModelBounds, default_weights, fkin6_model, get_acados_model_from_explicit_dynamics,
                  get_acados_ocp)
from .solver import BatchedOcpSolver
from .track import NUMBER_SPLINE_INTERVALS


class Controller(ABC):
    """``new_python/controller.py:134-159``."""

    id: str = "controller"

    @dataclass
    class Config:
        horizon_size: int
        sampling_time: float

    config: "Controller.Config"

    @abstractmethod
    def compute_control(self, x: np.ndarray, u: np.ndarray | None = None) -> np.ndarray:
        ...

    @property
    def sampling_time(self) -> float:
        return self.config.sampling_time


class StanleyController(Controller):
    """``python/main.py:99-163``; state arguments may be scalars or ``(B,)`` arrays."""

    id = "stanley"

    def __init__(self, k_P=90.0, k_I=20.0, k_offset=1.0, k_n=5.5, k_psi=1.8, k_kappa=1.0, T_max=500.0, delta_max=0.5,
                 dt=1 / 20):
        self.k_P, self.k_I, self.k_offset, self.k_n, self.k_psi, self.k_kappa = k_P, k_I, k_offset, k_n, k_psi, k_kappa
        self.T_max, self.delta_max, self.dt = T_max, delta_max, dt
        self.config = Controller.Config(horizon_size=0, sampling_time=dt)
        self.last_epsilons: deque = deque(maxlen=400)

    def compute_control(self, n, psi, v_x, v_x_ref, kappa_ref) -> np.ndarray:
        n, psi, v_x, kappa_ref = (np.asarray(a, dtype=np.float64) for a in (n, psi, v_x, kappa_ref))
        epsilon = v_x_ref - v_x
        u_T = self.k_P * epsilon
        if len(self.last_epsilons) > 1:
            acc = np.sum(self.last_epsilons, axis=0) - self.last_epsilons[0] - self.last_epsilons[-1]
            u_T = u_T + self.k_I * acc * self.dt
        self.last_epsilons.append(epsilon)
        u_delta = self.k_kappa * np.arctan(2 * np.tan(np.arcsin(kappa_ref * l_R)))
        u_delta = u_delta - self.k_psi * psi - np.arctan(self.k_n * n / (2.0 + v_x))
        return np.stack([np.clip(u_T, -self.T_max, self.T_max), np.clip(u_delta, -self.delta_max, self.delta_max)], axis=-1)


class IHM2Controller(Controller):
    """Batched counterpart of ``IHM2Controller`` (``python/main.py:166-334``)."""

    id = "ihm2"
    nx = NX
    nu = NU

    def __init__(
        self,
        s_ref: np.ndarray,
        kappa_ref: np.ndarray,
        Nf: int = 40,
        dt: float = 1 / 20,
        s_target: float = 40.0,
        n_max: float = 2.0,
        v_x_max: float = 31.0,
        T_max: float = 500.0,
        delta_max: float = 0.5,
        T_dot_max: float = 1e6,
        delta_dot_max: float = 1.0,
        a_lat_max: float = 5.0,
        q_s: float = 1.0, q_n: float = 1.0, q_psi: float = 1.0, q_v_x: float = 1.0, q_v_y: float = 1.0, q_r: float = 1.0,
        q_T: float = 1.0, q_delta: float = 100.0,
        q_s_f: float = 1000.0, q_n_f: float = 100.0, q_psi_f: float = 100.0, q_v_x_f: float = 1.0, q_v_y_f: float = 1.0,
        q_r_f: float = 1.0, q_T_f: float = 1.0, q_delta_f: float = 100.0,
        q_T_dot: float = 0.0, q_delta_dot: float = 500.0,
        *,
        batch_size: int = 1,
        track_id=None,
        device: int = 0,
        sim_method_num_steps: int = 25,
        integrator_type: str = "ERK",
        sim_integrator_type: str = "ERK",
        nlp_solver_type: str = "SQP_RTI",
        nlp_solver_max_iter: int = 1,
        globalization: str = "FIXED_STEP",
        nlp_tol: float = 1e-6,
        terminal_bounds: str = "reference",
        soft_state_bounds: tuple | None = None,
        track_widths=None,
        track_rows_penalty: tuple | None = (100.0, 100.0),
        lateral_acceleration_row: bool = False,
        recover_failed: bool = False,
    ) -> None:
        self.recover_failed = recover_failed
        self.Nf, self.dt, self.s_target, self.B = Nf, dt, s_target, int(batch_size)
        self.config = Controller.Config(horizon_size=Nf, sampling_time=dt)
        self.model_bounds = ModelBounds(n_max=n_max, v_x_min=0.0, v_x_max=v_x_max, T_max=T_max, delta_max=delta_max,
                                        T_dot_max=T_dot_max, delta_dot_max=delta_dot_max, a_lat_max=a_lat_max)
        s_ref = np.atleast_2d(np.asarray(s_ref, dtype=np.float64))
        model = get_acados_model_from_explicit_dynamics(
            name="ihm2_fkin6", continuous_model_fn=fkin6_model, x=self.nx, u=self.nu, p=2 * s_ref.shape[1])
        ocp = get_acados_ocp(model, Nf, n_max, v_x_max, T_max, delta_max, T_dot_max, delta_dot_max)
        if terminal_bounds == "stage":
            # quirk Q1: python/mpc.py:82-84 puts the T/delta limits on (v_y, r) at the terminal stage; with a plant
            # whose yaw rate is real (|r| = |v kappa| > delta_max) that box is infeasible.  "stage" repeats the
            # stage box [n, v_x, T, delta] at the terminal stage instead.
            c = ocp.constraints
            c.idxbx_e, c.lbx_e, c.ubx_e = c.idxbx.copy(), c.lbx.copy(), c.ubx.copy()
        elif terminal_bounds != "reference":
            raise ValueError("terminal_bounds must be 'reference' or 'stage'")
        if soft_state_bounds is not None:
            # not in the reference (python/mpc.py:58-90 has hard sides only): soften the path constraints on the
            # PLANT states -- the n and v_x boxes of every stage and the whole terminal box -- with the penalty
            # z s + 1/2 Z s^2, soft_state_bounds = (z, Z).  The actuator-state boxes, the input boxes and the rate rows
            # stay hard and can always be met, so the QP is feasible from any plant state.
            z_pen, Z_pen = (float(v) for v in soft_state_bounds)
            c = ocp.constraints
            pos = np.flatnonzero(np.isin(np.asarray(c.idxbx), (1, 3)))
            nsb, nsb_e = len(pos), len(c.idxbx_e)
            c.idxsbx, c.idxsbx_e = pos, np.arange(nsb_e)
            ocp.cost.zl = ocp.cost.zu = np.full(nsb, z_pen)
            ocp.cost.Zl = ocp.cost.Zu = np.full(nsb, Z_pen)
            ocp.cost.zl_e = ocp.cost.zu_e = np.full(nsb_e, z_pen)
            ocp.cost.Zl_e = ocp.cost.Zu_e = np.full(nsb_e, Z_pen)
        if track_widths is not None:
            # the nonlinear track-boundary rows of old/generate_acaods_interface.py:191-212,411-449 (footprint of the car
            # against the right / left width), soft with L1 + L2 weights (:380-395) unless track_rows_penalty is None
            model.con_h_expr = "track"
            c = ocp.constraints
            c.lh = c.lh_e = np.array([-1e3, -1e3])
            c.uh = c.uh_e = np.array([0.0, 0.0])
            if track_rows_penalty is not None:
                z_pen, Z_pen = (float(v) for v in track_rows_penalty)
                c.idxsh, c.idxsh_e = np.arange(2), np.arange(2)
                for name in ("zl", "zu"):
                    setattr(ocp.cost, name, np.concatenate([getattr(ocp.cost, name), np.full(2, z_pen)]))
                    setattr(ocp.cost, name + "_e", np.concatenate([getattr(ocp.cost, name + "_e"), np.full(2, z_pen)]))
                for name in ("Zl", "Zu"):
                    setattr(ocp.cost, name, np.concatenate([getattr(ocp.cost, name), np.full(2, Z_pen)]))
                    setattr(ocp.cost, name + "_e", np.concatenate([getattr(ocp.cost, name + "_e"), np.full(2, Z_pen)]))
            if lateral_acceleration_row:
                # the kinematic set's fifth row, |a_lat| <= ModelBounds.a_lat_max (old/generate_acaods_interface.py:198-209,
                # :424, :433), at the stages only; softened like the track rows (idxsh = all, :436)
                model.con_h_expr = "track+a_lat"
                c.lh = np.concatenate([c.lh, [-a_lat_max]]); c.uh = np.concatenate([c.uh, [a_lat_max]])
                if track_rows_penalty is not None:
                    c.idxsh = np.arange(3)
                    for name, v in (("zl", z_pen), ("zu", z_pen), ("Zl", Z_pen), ("Zu", Z_pen)):
                        setattr(ocp.cost, name, np.concatenate([getattr(ocp.cost, name), [v]]))
        elif lateral_acceleration_row:
            raise ValueError("lateral_acceleration_row comes with the track rows (track_widths): old/generate_acaods_interface.py:198-209")
        opts = AcadosOcpOptions()                      # python/main.py:227-238, with ERK x M for IRK
        opts.tf = Nf * dt
        opts.nlp_solver_type = nlp_solver_type
        opts.nlp_solver_max_iter = nlp_solver_max_iter
        opts.globalization = globalization              # python/main.py:237 asks "MERIT_BACKTRACKING" of its "SQP" solver
        opts.nlp_tol = nlp_tol
        opts.sim_method_num_steps = sim_method_num_steps
        # python/main.py:234-236: "IRK" = 4 Gauss-Legendre stages (acados' default collocation), sim_method_num_steps per interval;
        # sim_integrator_type: the plant steps behind this controller's device state (python/main.py:395-400: "IRK", GAUSS_RADAU_IIA)
        opts.integrator_type = integrator_type
        opts.sim_integrator_type, opts.sim_collocation_type = sim_integrator_type, "GAUSS_RADAU_IIA"
        ocp.solver_options = opts
        ocp.cost.W, ocp.cost.W_e = default_weights(q_s, q_n, q_psi, q_v_x, q_v_y, q_r, q_T, q_delta, q_s_f, q_n_f, q_psi_f,
                                                   q_v_x_f, q_v_y_f, q_r_f, q_T_f, q_delta_f, q_T_dot, q_delta_dot)
        self.solver = BatchedOcpSolver(ocp, self.B, s_ref, kappa_ref, track_id=track_id, device=device, track_widths=track_widths)
        # cold start of the prediction arrays (python/main.py:242-246)
        x_pred = np.zeros((self.B, Nf + 1, NX))
        x_pred[:, :, 0] = -6.0 + np.arange(Nf + 1) * dt
        u_pred = np.zeros((self.B, Nf, NU))
        u_pred[:, :, 0] = T_max
        self.solver.set_x(x_pred)
        self.solver.set_u(u_pred)
        self.last_status = np.zeros(self.B, dtype=np.int32)

    @classmethod
    def with_live_options(cls, s_ref, kappa_ref, *args, **kw) -> "IHM2Controller":
        """The controller exactly as the reference configures its solver (``python/main.py:227-238``): SQP with two iterations,
        MERIT_BACKTRACKING, IRK with four Gauss-Legendre stages and one step per shooting interval -- and its plants by Radau IIA
        (``:395-400``) unless ``sim_integrator_type`` says otherwise.  (The default constructor is the metric's SQP_RTI + RK4 x 25.)"""
        live = dict(nlp_solver_type="SQP", nlp_solver_max_iter=2, globalization="MERIT_BACKTRACKING", integrator_type="IRK",
                    sim_method_num_steps=1, sim_integrator_type="IRK")
        live.update(kw)
        return cls(s_ref, kappa_ref, *args, **live)

    # -- predictions of the last solve (python/main.py:331-332) --
    @property
    def x_pred(self) -> np.ndarray:
        return self.solver.get_x()

    @property
    def u_pred(self) -> np.ndarray:
        return self.solver.get_u()

    def compute_control(self, x: np.ndarray, u: np.ndarray | None = None):
        """``x``: ``(8,)`` for a batch of one or ``(B, 8)``.  Returns ``u0`` of the same leading shape; like the
        reference (``python/main.py:326-328``) a batch of one returns ``None`` when the status is not in {0, 2};
        for a batch the failed rows are NaN and ``last_status`` holds the codes."""
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        # solver.set(0, "lbx"/"ubx", x), yref ramp + warm-start shift (on device), solve, solver.get(0, "u"): one call, one wait
        u0, self.last_status = self.solver.compute_control(x.reshape(self.B, NX), self.s_target)
        if self.recover_failed:
            # not in the reference (its loop stops at the first bad status, python/main.py:326-328): give the failed instances a
            # fresh rollout as the next warm start instead of the iterate that made their QP infeasible
            self.solver.reinit_failed()
        bad = ~np.isin(self.last_status, (0, 2))
        if single:
            return None if bad[0] else u0[0]
        u0[bad] = np.nan
        return u0

    def warm_start(self, x0: np.ndarray, v_ref_scale: float = 1.0) -> None:
        """Replace the cold-start prediction (a car at rest at s = -6, ``python/main.py:242-246``) by a rollout of the
        model from ``x0`` under Stanley feedback -- for batches that start anywhere on the track at speed."""
        self.solver.set_x0(np.asarray(x0, dtype=np.float64).reshape(self.B, NX))
        self.solver.init_guess(v_ref_scale)

    def compute_control_device(self):
        """Zero-copy step for closed loops whose state already lives on the device (``sim_advance``)."""
        self.solver.prepare_step(self.s_target)
        self.solver.solve_async()
