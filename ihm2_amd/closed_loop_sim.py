"""Model-in-the-loop closed loop, batched on the GPU (BASELINE.json config 5).

Semantics of the reference's MiL loop ``main()`` (``python/main.py:438-517``): start at ``s = -6``, every
control step call the controller, advance the plant in Frenet coordinates with the same track table,
abort an instance on NaN, stop it one metre after a full lap, report lap time / mean speed / controller
run time (``:561-575``).  Plant variants ``SimModelVariant`` (``python/main.py:337-341``,
``new_python/closed_loop_sim.py:27-31``): KIN6, DYN6, KIN6_DYN6 (speed switch ``v^2 sin(beta)/l_R <= 3``,
``python/main.py:482-489``), DYN10 (the 15-state plant with wheel speeds under the Stanley controller, ``python/main.py:459-465,490-502``:
:class:`Dyn10Simulator`, :func:`run_closed_loop_dyn10`).  The reference integrates the plants with IRK Radau-IIA, 4 stages x 100 steps
(``python/main.py:395-400``): the DYN10 plant does the same here (from rest), the 8-state plants of the batched NMPC loop are RK4 with
``M_sim`` sub-steps (default 100; ``ihm2mpc_sim_step`` also offers Radau IIA).

Names kept from the refactor skeleton: ``SimulatorConfig`` (sic ``colloaction_type``), ``Simulator``,
``MultiModelSimulator`` (``new_python/simulator.py:25-62``), ``closed_loop`` with its argument list
(``new_python/closed_loop_sim.py:34-42``).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from enum import Enum, auto
from typing import Callable

import numpy as np

from .controller import IHM2Controller
from .track import MotionPlan, track_table


class SimModelVariant(Enum):
    KIN6 = auto()
    DYN6 = auto()
    KIN6_DYN6 = auto()
    DYN10 = auto()          # 15 states, Stanley controller: Dyn10Simulator / run_closed_loop_dyn10 below
    DYN6U = auto()          # DYN6 / KIN6_DYN6 with un-crossed slip angles (named deviation from quirk Q3)
    KIN6_DYN6U = auto()


    # Cartesian plants of the ROS simulation node (src/ihm2/src/sim_node.cpp:197-257; python/models.py:168-229, 310-452)
    CART_KIN6 = auto()
    CART_DYN6 = auto()
    CART_KIN6_DYN6 = auto()   # the node's speed switch (v < v_dyn) and its "no reversing" clamp


_PLANT_CODE = {SimModelVariant.KIN6: 0, SimModelVariant.DYN6: 1, SimModelVariant.KIN6_DYN6: -1, SimModelVariant.DYN6U: 2,
               SimModelVariant.KIN6_DYN6U: -2}
_CART_CODE = {SimModelVariant.CART_KIN6: 3, SimModelVariant.CART_DYN6: 4, SimModelVariant.CART_KIN6_DYN6: -3}


@dataclass
class SimulatorConfig:
    """``new_python/simulator.py:25-40`` (sic ``colloaction_type``).  ``integrator_type``: ``None`` = whatever the controller's device state
    was created with (``IHM2Controller(sim_integrator_type=...)``: "ERK" = RK4 x num_steps, "IRK" = Radau IIA x num_steps as
    ``python/main.py:395-400``); a value is checked against it."""

    sampling_time: float
    integrator_type: str | None = None
    colloaction_type: str = "GAUSS_RADAU_IIA"
    num_steps: int = 100


class Simulator:
    """Batched plant on the device that already holds the controller's state (``new_python/simulator.py:42-47``)."""

    def __init__(self, controller: IHM2Controller, config: SimulatorConfig, variant: SimModelVariant = SimModelVariant.KIN6_DYN6):
        if variant not in _PLANT_CODE:
            raise NotImplementedError(f"plant variant {variant.name}")
        have = controller.solver.ocp.solver_options.sim_integrator_type
        if config.integrator_type is not None and config.integrator_type != have:
            raise ValueError(f'SimulatorConfig.integrator_type = {config.integrator_type!r}, but the controller\'s device state integrates its plants with '
                             f'{have!r}: create the controller with sim_integrator_type={config.integrator_type!r}')
        self.controller, self.config, self.variant = controller, config, variant

    def simulate(self, x: np.ndarray, u: np.ndarray) -> np.ndarray:
        """Host arrays in, host arrays out: ``x`` (B,8), ``u`` (B,2) -> next state (B,8)."""
        return self.controller.solver.sim_step(x, u, model=_PLANT_CODE[self.variant], M_sim=self.config.num_steps)

    def advance(self) -> None:
        """x0 <- plant(x0, u0 of the last solve) without leaving the device."""
        self.controller.solver.sim_advance(model=_PLANT_CODE[self.variant], M_sim=self.config.num_steps)


class CartesianSimulator:
    """Plant in Cartesian coordinates as the ROS stack runs it: ``simulate(x_cart, u)`` advances ``(X, Y, phi, v_x, v_y, r, T,
    delta)`` by one control period in 100 Hz plant steps (``sim_node.cpp:197-257``), ``project`` gives the controller its
    Frenet state (``Track::project`` + ``mpc_control_node.cpp:142-157``).  The signature of ``new_python/simulator.py:25-40``
    (``simulate(X, Y, phi, v_x, v_y, r, T, delta)``) is the unpacked form of ``simulate``."""

    def __init__(self, controller: IHM2Controller, plan: MotionPlan, variant: SimModelVariant = SimModelVariant.CART_KIN6_DYN6,
                 plant_dt: float = 0.01, num_steps: int = 10, v_dyn: float = 3.0, s_tol: float = 2.0):
        if variant not in _CART_CODE:
            raise NotImplementedError(f"Cartesian plant variant {variant.name}")
        self.controller, self.variant = controller, variant
        self.plant_dt, self.num_steps, self.v_dyn, self.s_tol = plant_dt, num_steps, v_dyn, s_tol
        self.n_plant_steps = max(1, int(round(controller.dt / plant_dt)))
        controller.solver.set_track_geometry(plan.X_ref, plan.Y_ref, plan.phi_ref)

    def simulate(self, x_cart: np.ndarray, u: np.ndarray) -> np.ndarray:
        return self.controller.solver.sim_step_cart(x_cart, u, model=_CART_CODE[self.variant], M_sim=self.num_steps, dt_sim=self.plant_dt,
                                                    n_steps=self.n_plant_steps, v_dyn=self.v_dyn)

    def project(self, x_cart: np.ndarray, s_guess: np.ndarray):
        return self.controller.solver.project(x_cart, s_guess, self.s_tol)


def frenet_to_cartesian(plan: MotionPlan, x_frenet: np.ndarray) -> np.ndarray:
    """Cartesian state of a Frenet state on the (piecewise linear) centre line: position = point at s + n * normal."""
    xf = np.atleast_2d(x_frenet)
    s = xf[:, 0]
    Xc = np.interp(s, plan.s_ref, plan.X_ref); Yc = np.interp(s, plan.s_ref, plan.Y_ref)
    i = np.clip(np.searchsorted(plan.s_ref, s, side="right") - 1, 0, len(plan.s_ref) - 2)
    th = np.arctan2(plan.Y_ref[i + 1] - plan.Y_ref[i], plan.X_ref[i + 1] - plan.X_ref[i])      # heading of the segment
    out = xf.copy()
    out[:, 0] = Xc - xf[:, 1] * np.sin(th)
    out[:, 1] = Yc + xf[:, 1] * np.cos(th)
    out[:, 2] = np.interp(s, plan.s_ref, np.unwrap(plan.phi_ref)) + xf[:, 2]
    return out


def run_closed_loop_cartesian(controller: IHM2Controller, simulator: CartesianSimulator, x_cart0: np.ndarray, s_guess0: np.ndarray,
                              n_steps: int, lap_length: float | None = None) -> ClosedLoopResult:
    """The ROS loop for a batch: project -> controller -> Cartesian plant.  ``ClosedLoopResult.x`` holds the FRENET states."""
    B = controller.B
    xc = np.asarray(x_cart0, dtype=np.float64).reshape(B, 8).copy()
    sg = np.asarray(s_guess0, dtype=np.float64).reshape(B).copy()
    xs, us, sts, runtimes, alive_hist = [], [], [], [], []
    alive = np.ones(B, dtype=bool); finished = np.zeros(B, dtype=bool); lap_time = np.full(B, np.nan)
    progress = np.zeros(B); s_prev = None
    for i in range(n_steps):
        alive_hist.append(alive.copy())
        t0 = time.perf_counter()
        xf, sg_new = simulator.project(xc, sg)
        u = controller.compute_control(xf)
        runtimes.append((time.perf_counter() - t0) * 1e3)
        st = controller.last_status.copy()
        alive &= ~(alive & ~np.isin(st, (0, 2)))
        u = np.where(alive[:, None], np.nan_to_num(u), 0.0)
        xn = simulator.simulate(xc, u)
        alive &= ~(alive & np.any(np.isnan(xn), axis=1))
        if s_prev is not None:
            ds = xf[:, 0] - s_prev
            if lap_length is not None:
                ds = np.where(ds < -0.5 * lap_length, ds + lap_length, ds)     # s_guess wraps with fmod(., lap length)
            progress += np.where(alive, ds, 0.0)
        s_prev = xf[:, 0].copy()
        xc = np.where(alive[:, None], xn, xc); sg = np.where(alive, sg_new, sg)
        if lap_length is not None:
            done = alive & (progress > lap_length)
            lap_time[done] = (i + 1) * controller.dt
            finished |= done; alive &= ~done
        xs.append(xf.copy()); us.append(u); sts.append(st)
        if not alive.any():
            break
    xs.append(xs[-1])
    res = ClosedLoopResult(np.array(xs), np.array(us), np.array(sts), alive, finished, lap_time, runtimes, np.array(alive_hist))
    res.progress = progress
    return res


class Dyn10Simulator:
    """The DYN10 plant of ``python/main.py:395-428,490-502``: ``fdyn10`` (15 states: Frenet pose, velocities, four wheel speeds, four
    wheel torques, steering) under ``AcadosSimOpts(T = dt, num_stages = 4, num_steps = 100, "IRK", "GAUSS_RADAU_IIA")``, for a batch of cars.
    ``simulate(x (B,15), u (B,2))`` splits the torque command over the wheels as the reference's loop does."""

    def __init__(self, plan: MotionPlan, batch_size: int = 1, sampling_time: float = 0.05, num_steps: int = 100, device: int = 0,
                 car: "CarParams | None" = None):
        from . import ocp as O
        from .car_params import default_car_params
        from .sim import AcadosSimOpts, generate_sim_solver

        self.car = default_car_params() if car is None else car
        self.B, self.dt = int(batch_size), float(sampling_time)
        model = O.get_acados_model_from_implicit_dynamics("ihm2_fdyn10", O.fdyn10_model, 15, 5, 2 * plan.s_ref.size)
        opts = AcadosSimOpts(T=self.dt, num_stages=4, num_steps=num_steps, integrator_type="IRK", collocation_type="GAUSS_RADAU_IIA")
        self.sim = generate_sim_solver(model, opts, "", generate=False, build=False, batch_size=self.B, device=device)
        self.sim.set("p", np.append(plan.s_ref, plan.kappa_ref))                          # python/main.py:432-435

    def initial_state(self, s0: float = -6.0) -> np.ndarray:
        """``python/main.py:438-441``: the car stands at ``s = s0`` on the centre line."""
        from .car_params import CarState

        return np.tile(CarState().to_frenet_dyn10(s0, 0.0, 0.0), (self.B, 1))

    def wheel_commands(self, u: np.ndarray) -> np.ndarray:
        """``python/main.py:492-500``: a quarter of the torque command on every wheel (within the wheel's limit), the steering command as is."""
        u = np.asarray(u, dtype=np.float64).reshape(self.B, 2)
        tau = np.clip(0.25 * u[:, 0], -self.car.actuator_params.wheel_torque_max, self.car.actuator_params.wheel_torque_max)
        return np.column_stack([tau, tau, tau, tau, u[:, 1]])

    def simulate(self, x: np.ndarray, u: np.ndarray) -> np.ndarray:
        self.sim.set("x", np.asarray(x, dtype=np.float64).reshape(self.B, 15)); self.sim.set("u", self.wheel_commands(u))
        self.sim.solve()                                     # a car whose step fails comes back as a NaN row (the loop stops that car)
        return np.asarray(self.sim.get("x")).reshape(self.B, 15)

    def free(self) -> None:
        self.sim.free()


def run_closed_loop_dyn10(plan: MotionPlan, n_steps: int = 501, batch_size: int = 1, v_x_ref=5.0, dt: float = 0.05, device: int = 0,
                          interation_end_callback: Callable | None = None) -> "ClosedLoopResult":
    """The reference's MiL loop with ``sim_model_variant = DYN10`` as it runs (``python/main.py:438-517``): from rest at ``s = -6``, the
    Stanley controller (``:459-465``; the NMPC call at ``:457`` is commented out there and would be handed the 15-state vector) tracks
    ``v_x_ref`` (a scalar or one value per car), the plant is ``fdyn10`` under Radau IIA; a NaN state stops a car, ``s > lap + 1`` ends its lap."""
    from .controller import StanleyController

    B = int(batch_size)
    sim = Dyn10Simulator(plan, B, dt, device=device)
    ctrl = StanleyController(dt=dt)
    x = sim.initial_state()
    v_ref = np.broadcast_to(np.asarray(v_x_ref, dtype=np.float64), (B,))
    xs, us, sts, runtimes, alive_hist = [x.copy()], [], [], [], []
    alive = np.ones(B, dtype=bool); finished = np.zeros(B, dtype=bool); lap_time = np.full(B, np.nan)
    for i in range(n_steps):
        alive_hist.append(alive.copy())
        t0 = time.perf_counter()
        u = ctrl.compute_control(n=x[:, 1], psi=x[:, 2], v_x=x[:, 3], v_x_ref=v_ref, kappa_ref=np.interp(x[:, 0], plan.s_ref, plan.kappa_ref))
        runtimes.append((time.perf_counter() - t0) * 1e3)
        u = np.where(alive[:, None], u, 0.0)
        xn = sim.simulate(x, u)
        nan = alive & np.any(np.isnan(xn), axis=1)          # python/main.py:503-504
        alive &= ~nan
        x = np.where(alive[:, None], xn, x)
        done = alive & (x[:, 0] > plan.lap_length + 1.0)    # python/main.py:514-517
        lap_time[done] = (i + 1) * dt
        finished |= done; alive &= ~done
        xs.append(x.copy()); us.append(u); sts.append(np.where(nan, 1, 0).astype(np.int32))
        if interation_end_callback is not None:
            interation_end_callback(i, x, u, sts[-1])
        if not alive.any():
            break
    sim.free()
    return ClosedLoopResult(np.array(xs), np.array(us), np.array(sts), alive, finished, lap_time, runtimes, np.array(alive_hist))


class MultiModelSimulator(Simulator):
    """``new_python/simulator.py:54-62``: the kinematic/dynamic switch is evaluated per instance on the device."""

    def __init__(self, controller: IHM2Controller, config: SimulatorConfig, condition: Callable | None = None):
        super().__init__(controller, config, SimModelVariant.KIN6_DYN6)
        self.condition = condition


@dataclass
class ClosedLoopResult:
    x: np.ndarray            # (steps+1, B, 8)
    u: np.ndarray            # (steps, B, 2)
    status: np.ndarray       # (steps, B)
    alive: np.ndarray        # (B,) still running at the end
    finished: np.ndarray     # (B,) completed a lap
    lap_time: np.ndarray     # (B,) seconds (nan if not finished)
    runtimes_ms: list = field(default_factory=list)
    alive_history: np.ndarray | None = None   # (steps, B) alive at the start of each step

    def stats(self) -> dict:
        v = np.hypot(self.x[..., 3], self.x[..., 4])
        return {"laps_finished": int(self.finished.sum()), "failed": int((~self.alive & ~self.finished).sum()),
                "mean_speed": float(np.nanmean(v)), "mean_runtime_ms": float(np.mean(self.runtimes_ms)),
                "max_runtime_ms": float(np.max(self.runtimes_ms)),
                "control_steps_per_s": float(self.u.shape[0] * self.u.shape[1] / (np.sum(self.runtimes_ms) * 1e-3))}


def run_closed_loop(controller: IHM2Controller, simulator: Simulator, x0: np.ndarray, n_steps: int, lap_length: float | None = None,
                    interation_end_callback: Callable | None = None) -> ClosedLoopResult:
    """``python/main.py:448-517`` for a batch.  Instances that fail (status not in {0,2}, NaN) or finish the lap are
    frozen: they keep their last state and a zero input, as the reference stops its single loop."""
    B = controller.B
    x = np.asarray(x0, dtype=np.float64).reshape(B, 8).copy()
    xs, us, sts, runtimes = [x.copy()], [], [], []
    alive = np.ones(B, dtype=bool)
    finished = np.zeros(B, dtype=bool)
    lap_time = np.full(B, np.nan)
    alive_hist = []
    for i in range(n_steps):
        alive_hist.append(alive.copy())
        t0 = time.perf_counter()
        u = controller.compute_control(x)
        runtimes.append((time.perf_counter() - t0) * 1e3)
        st = controller.last_status.copy()
        bad = alive & ~np.isin(st, (0, 2))
        alive &= ~bad
        u = np.where(alive[:, None], np.nan_to_num(u), 0.0)
        xn = simulator.simulate(x, u)
        nan = alive & np.any(np.isnan(xn), axis=1)          # python/main.py:503-504
        alive &= ~nan
        x = np.where(alive[:, None], xn, x)
        if lap_length is not None:
            done = alive & (x[:, 0] > lap_length + 1.0)     # python/main.py:514-517
            lap_time[done] = (i + 1) * controller.dt
            finished |= done
            alive &= ~done
        xs.append(x.copy()); us.append(u); sts.append(st)
        if interation_end_callback is not None:
            interation_end_callback(i, x, u, st)
        if not alive.any():
            break
    return ClosedLoopResult(np.array(xs), np.array(us), np.array(sts), alive, finished, lap_time, runtimes, np.array(alive_hist))


def run_closed_loop_device(controller: IHM2Controller, simulator: Simulator, x0: np.ndarray, n_steps: int, lap_length: float | None = None
                           ) -> ClosedLoopResult:
    """Same loop as :func:`run_closed_loop`, with the state resident on the device: one ``ihm2mpc_step`` per control period
    (plant beside the linearisation), and per step only ``x0`` (B,8), ``u0`` (B,2) and the status (B,) cross the boundary for
    the bookkeeping.  Frozen instances (failed / finished) are masked out of the plant (``ihm2mpc_set_active``)."""
    B, s = controller.B, controller.solver
    code = _PLANT_CODE[simulator.variant]
    x = np.asarray(x0, dtype=np.float64).reshape(B, 8).copy()
    xs, us, sts, runtimes, alive_hist = [x.copy()], [], [], [], []
    alive = np.ones(B, dtype=bool); finished = np.zeros(B, dtype=bool); lap_time = np.full(B, np.nan)
    t0 = time.perf_counter()
    s.set_active(None)
    s.set_x0(x); s.prepare_step(controller.s_target); st = s.solve(); u = s.get_u0()          # control for the initial state
    runtimes.append((time.perf_counter() - t0) * 1e3)
    mask_dirty = False
    for i in range(n_steps):
        alive_hist.append(alive.copy())
        bad = alive & ~np.isin(st, (0, 2))
        if bad.any():
            alive &= ~bad; mask_dirty = True
        us.append(np.where(alive[:, None], np.nan_to_num(u), 0.0)); sts.append(st.copy())
        if not alive.any():
            xs.append(x.copy())
            break
        if mask_dirty:
            s.set_active(alive.astype(np.int32)); mask_dirty = False
        t0 = time.perf_counter()
        s.step(controller.s_target, model=code, M_sim=simulator.config.num_steps)     # x <- plant(x, u); u <- NMPC(x)
        xn = s.get_x0(); u = s.get_u0(); st = s.get_status()
        runtimes.append((time.perf_counter() - t0) * 1e3)
        nan = alive & np.any(np.isnan(xn), axis=1)
        if nan.any():
            alive &= ~nan; mask_dirty = True
        x = np.where(alive[:, None], xn, x)
        if lap_length is not None:
            done = alive & (x[:, 0] > lap_length + 1.0)
            if done.any():
                lap_time[done] = (i + 1) * controller.dt
                finished |= done; alive &= ~done; mask_dirty = True
        xs.append(x.copy())
    controller.last_status = st
    s.set_active(None)
    return ClosedLoopResult(np.array(xs), np.array(us), np.array(sts), alive, finished, lap_time, runtimes, np.array(alive_hist))


def run_closed_loop_persistent(controller: IHM2Controller, simulator: Simulator, x0: np.ndarray, n_steps: int, lap_length: float | None = None
                               ) -> ClosedLoopResult:
    """Same loop as :func:`run_closed_loop_device` in ONE launch (``ihm2mpc_run_steps``): every car runs its control periods back
    to back on its own wavefront, the freezing rules (failed solve, NaN plant state, lap done) are applied on the device, and the
    histories come back once at the end.  For batches that fit the device at once (4 cars per compute unit) and the reference's
    OCP; otherwise the solver reports it and :func:`run_closed_loop_device` is the loop to use."""
    B, s = controller.B, controller.solver
    code = _PLANT_CODE[simulator.variant]
    x = np.asarray(x0, dtype=np.float64).reshape(B, 8).copy()
    t0 = time.perf_counter()
    s.set_active(None)
    s.set_x0(x); s.prepare_step(controller.s_target); st0 = s.solve(); u_first = s.get_u0()     # control for the initial state
    h = s.run_steps(controller.s_target, n_steps, model=code, M_sim=simulator.config.num_steps, freeze=True,
                    lap_stop=np.inf if lap_length is None else lap_length + 1.0, u0_hist=True, x0_hist=True, status_hist=True)
    wall_ms = (time.perf_counter() - t0) * 1e3
    us_raw = np.concatenate([u_first[None], h["u0"][:-1]]); sts = np.concatenate([st0[None], h["status"][:-1]])
    xs = np.concatenate([x[None], h["x0"]])
    # the bookkeeping of run_closed_loop_device, replayed on the histories (the device applied the same rules)
    alive = np.ones(B, dtype=bool); finished = np.zeros(B, dtype=bool); lap_time = np.full(B, np.nan)
    alive_hist, us = [], []
    n_done = n_steps
    for i in range(n_steps):
        alive_hist.append(alive.copy())
        alive &= np.isin(sts[i], (0, 2))
        us.append(np.where(alive[:, None], np.nan_to_num(us_raw[i]), 0.0))
        if not alive.any():
            n_done = i + 1
            break
        moved = np.any(xs[i + 1] != xs[i], axis=1)
        alive &= moved | ~alive_hist[-1]          # a NaN plant state froze the car where it was
        if lap_length is not None:
            done = alive & (xs[i + 1][:, 0] > lap_length + 1.0)
            lap_time[done] = (i + 1) * controller.dt
            finished |= done; alive &= ~done
    controller.last_status = s.get_status()
    s.set_active(None)
    return ClosedLoopResult(xs[:n_done + 1], np.array(us), sts[:n_done], alive, finished, lap_time, [wall_ms / max(n_done, 1)] * n_done,
                            np.array(alive_hist))


def closed_loop(track_data: str | MotionPlan, simulator_type=SimModelVariant.KIN6_DYN6, motion_planner_type: type | None = None,
                motion_tracker_type: type = IHM2Controller, low_level_controller_type: type | None = None,
                interation_end_callback: Callable | None = None, cleanup_callback: Callable | None = None, *,
                batch_size: int = 1, n_steps: int = 501, x0: np.ndarray | None = None, **controller_kwargs) -> ClosedLoopResult:
    """Argument list of ``new_python/closed_loop_sim.py:34-42`` (a pseudo-code skeleton there).  ``track_data`` is a
    track name or a tripled motion plan; the motion tracker is the NMPC controller; there is no separate motion
    planner / low-level controller in the reference's loop (``python/main.py:448-517``)."""
    plan = track_table(track_data) if isinstance(track_data, str) else track_data
    controller = motion_tracker_type(plan.s_ref, plan.kappa_ref, batch_size=batch_size, **controller_kwargs)
    sim = Simulator(controller, SimulatorConfig(sampling_time=controller.dt), simulator_type)
    if x0 is None:
        x0 = np.zeros((batch_size, 8))
        x0[:, 0] = -6.0                                     # python/main.py:438-441
    res = run_closed_loop(controller, sim, x0, n_steps, plan.lap_length, interation_end_callback)
    if cleanup_callback is not None:
        cleanup_callback(res)
    return res
