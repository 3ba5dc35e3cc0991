"""``python/sim.py`` on the device: ``AcadosSimOpts`` / ``AcadosSim`` / ``AcadosSimSolver``-shaped plant integrator and
``generate_sim_solver(model, opts, gen_code_dir, **kwargs)`` over the C ABI's ``ihm2mpc_sim_step``.

The reference builds one ``AcadosSimSolver`` per plant model (``python/main.py:395-428``: IRK, GAUSS_RADAU_IIA, 4 stages, 100 steps,
``T = dt``), hands it the track as parameter vector ``p = [s_ref; kappa_ref]`` (``:432-435``) and calls ``simulate(x, u)`` once per
control period (``:476-502``).  Here the same object integrates a BATCH: ``x`` may be ``(8,)`` (the reference's call, returns ``(8,)``)
or ``(B, 8)``.  Fields of ``set`` / ``get`` as used by the reference and by ``old/scripts/gen_sim.py:599-604``: ``"x"``, ``"u"``,
``"p"``, ``"T"`` in, ``"x"`` (the state after ``solve``), ``"CPUtime"`` / ``"time_tot"`` (seconds of the last ``solve``) out.
No code generation: ``gen_code_dir`` and the ``generate`` / ``build`` keywords are accepted and ignored.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from .ocp import NU, NX, AcadosModel, AcadosOcp, AcadosOcpOptions, default_weights, fkin6_model, get_acados_model_from_explicit_dynamics, get_acados_ocp


@dataclass
class AcadosSimOpts:
    """``acados_template.AcadosSimOpts`` (the attributes the reference sets, ``python/sim.py:28-33``, ``python/main.py:395-400``)."""

    T: float = 0.01
    num_stages: int = 4
    num_steps: int = 1
    integrator_type: str = "IRK"
    collocation_type: str = "GAUSS_LEGENDRE"
    newton_iter: int = 3


@dataclass
class AcadosSim:
    """``acados_template.AcadosSim``."""

    model: AcadosModel = field(default_factory=AcadosModel)
    solver_options: AcadosSimOpts = field(default_factory=AcadosSimOpts)
    code_export_directory: str = ""
    parameter_values: np.ndarray | None = None


class AcadosSimSolver:
    """``acados_template.AcadosSimSolver`` for the models of this build (``fkin6``, ``fdyn6``, ``fdyn6u``, ``fdyn10``), batched."""

    def __init__(self, sim: AcadosSim, json_file: str | None = None, verbose: bool = False, batch_size: int = 1, device: int = 0, **kwargs):
        o = sim.solver_options
        if o.num_stages != 4:
            raise ValueError("4 stages: the classical RK4 (ERK) or 4-stage collocation (IRK), python/main.py:397")
        if o.integrator_type not in ("ERK", "IRK"):
            raise ValueError(f"integrator_type {o.integrator_type!r}")
        self.sim, self.B, self.device = sim, int(batch_size), int(device)
        self.model_id = sim.model.model_id
        if o.integrator_type == "IRK" and o.newton_iter != 3 and sim.model.kind != "fdyn10":
            raise ValueError("the IRK integrator runs acados' default of 3 Newton iterations per step")
        # fdyn10 (python/models.py:609-801; the DYN10 plant of python/main.py:490-502): 15 states, 5 inputs; "IRK" = the reference's
        # Radau IIA x num_steps solved to convergence (usable from rest, python/main.py:438-441), "ERK" = RK4 x num_steps (moving cars)
        self.dyn10 = sim.model.kind == "fdyn10"
        self.nx, self.nu = (15, 5) if self.dyn10 else (NX, NU)
        self._T = float(o.T)
        self._p = None if sim.parameter_values is None else np.asarray(sim.parameter_values, dtype=float).ravel()
        self._x = np.zeros((self.B, self.nx)); self._u = np.zeros((self.B, self.nu)); self._xn = np.zeros((self.B, self.nx))
        self._time = 0.0
        self._solver = None

    # -- the device object is made when T and p are known (both may be set after construction, python/main.py:432-435) --
    def _backend(self):
        if self._solver is not None:
            return self._solver
        from .solver import BatchedOcpSolver

        p = self._p
        if p is None or p.size < 4 or p.size % 2 or not np.all(np.diff(p[: p.size // 2]) > 0):
            raise ValueError('set("p", [s_ref; kappa_ref]) first: the track table the models interpolate (python/main.py:432-435)')
        o = self.sim.solver_options
        carrier = get_acados_model_from_explicit_dynamics("carrier", fkin6_model, NX, NU, p.size) if self.dyn10 else self.sim.model
        ocp: AcadosOcp = get_acados_ocp(carrier, 2, 1e3, 1e3, 1e6, 10.0, 1e9, 1e9)      # a carrier for the plant: never solved
        ocp.cost.W, ocp.cost.W_e = default_weights()
        so: AcadosOcpOptions = ocp.solver_options
        so.tf = 2 * self._T
        so.sim_integrator_type, so.sim_collocation_type = o.integrator_type, o.collocation_type
        # sub-steps of the (never solved) carrier OCP: RK4 stays stable on the 1 ms torque lag whatever T is, so that
        # ihm2mpc_create's stability check cannot refuse a plant that would never run RK4 on it
        so.sim_method_num_steps = max(25, int(np.ceil(self._T / 2e-3)))
        n = p.size // 2
        self._solver = BatchedOcpSolver(ocp, self.B, p[:n], p[n:], device=self.device)
        return self._solver

    def set(self, field_: str, value) -> None:
        v = np.asarray(value, dtype=float)
        if field_ == "x":
            self._x[...] = v.reshape(-1, self.nx) if v.ndim > 1 else v
        elif field_ == "u":
            self._u[...] = v.reshape(-1, self.nu) if v.ndim > 1 else v
        elif field_ == "p":
            self._p = v.ravel().copy()
            if self._solver is not None:
                self._solver.free(); self._solver = None
        elif field_ == "T":
            self._T = float(v)
            if self._solver is not None:
                self._solver.free(); self._solver = None
        else:
            raise ValueError(f"unknown field {field_!r}: x, u, p, T")

    def solve(self) -> int:
        """One integration over ``T``; status 0, or 1 if a state came back non-finite (acados: ACADOS_FAILURE)."""
        s = self._backend()
        t0 = time.perf_counter()
        if self.dyn10:
            self._xn = s.sim_step_dyn10(self._x, self._u, M_sim=int(self.sim.solver_options.num_steps))
        else:
            self._xn = s.sim_step(self._x, self._u, model=self.model_id, M_sim=int(self.sim.solver_options.num_steps))
        self._time = time.perf_counter() - t0
        return 0 if np.all(np.isfinite(self._xn)) else 1

    def get(self, field_: str):
        if field_ in ("x", "xn"):
            return self._xn[0].copy() if self.B == 1 else self._xn.copy()
        if field_ in ("CPUtime", "time_tot"):
            return self._time
        raise ValueError(f"unknown field {field_!r}: x, CPUtime")

    def simulate(self, x=None, u=None, p=None):
        """``AcadosSimSolver.simulate`` (``python/main.py:479-491``): set what is given, solve, return the next state."""
        if p is not None:
            self.set("p", p)
        if x is not None:
            self.set("x", x)
        if u is not None:
            self.set("u", u)
        status = self.solve()
        if status != 0:
            raise RuntimeError(f"simulation failed with status {status}")
        return self.get("x")

    def free(self) -> None:
        if self._solver is not None:
            self._solver.free(); self._solver = None


def generate_sim_solver(model: AcadosModel, opts: AcadosSimOpts, gen_code_dir: str, **kwargs) -> AcadosSimSolver:
    """``python/sim.py:9-25``.  ``generate=`` / ``build=`` (code generation switches of acados) are accepted and ignored;
    ``batch_size=`` and ``device=`` select the batch this plant integrates."""
    sim = AcadosSim()
    sim.model = model
    sim.solver_options = opts
    sim.code_export_directory = gen_code_dir + "/ihm2_fkin6_sim_gen_code"
    sim.parameter_values = None
    kwargs.pop("generate", None); kwargs.pop("build", None)
    return AcadosSimSolver(sim, json_file=gen_code_dir + "/ihm2_fkin6_sim.json", verbose=False, **kwargs)


default_sim_solver_opts = AcadosSimOpts()
default_sim_solver_opts.T = 0.01
default_sim_solver_opts.num_stages = 4
default_sim_solver_opts.num_steps = 1
default_sim_solver_opts.integrator_type = "IRK"
default_sim_solver_opts.collocation_type = "GAUSS_RADAU_IIA"
