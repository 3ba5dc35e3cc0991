"""OCP description of the path-parametric bicycle NMPC (host side, NumPy only).

Mirrors the reference's solver factory module ``python/mpc.py`` (``ModelBounds`` :18-26,
``get_acados_ocp`` :29-101, ``get_acados_solver`` :104-113) and the model wrappers of
``python/models.py:809-843``.  The reference fills an ``acados_template.AcadosOcp``; here the same
attribute tree (``ocp.dims``, ``ocp.cost``, ``ocp.constraints``, ``ocp.solver_options``) is a set of
plain dataclasses, and :meth:`AcadosOcp.flatten` turns it into the dense arrays the C-ABI
(``include/ihm2mpc.h``) takes.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .constants import NC, NG, NH, NLAM, NU, NX, NY, car_length, car_width, t_delta, t_T

INF = float("inf")

MODEL_FKIN6 = 0
MODEL_FDYN6 = 1
MODEL_FDYN6U = 2     # fdyn6 with un-crossed slip angles (include/ihm2mpc.h: IHM2MPC_MODEL_FDYN6U); not in the reference
MODEL_FDYN10 = 10      # plant only (15 states, 5 inputs): no OCP model id in the C ABI, its own entry point ihm2mpc_sim_step_dyn10
_MODEL_IDS = {"fkin6": MODEL_FKIN6, "fdyn6": MODEL_FDYN6, "fdyn6u": MODEL_FDYN6U, "fdyn10": MODEL_FDYN10}
INTEG_RK4 = 0
INTEG_IRK_GL4 = 1      # IRK, GAUSS_LEGENDRE, 4 stages: acados' default collocation (python/main.py:234-236)
INTEG_IRK_RADAU4 = 2   # IRK, GAUSS_RADAU_IIA, 4 stages (python/main.py:395-400, python/sim.py:28-33)


def integrator_code(integrator_type: str, collocation_type: str = "GAUSS_LEGENDRE") -> int:
    """``AcadosOcpOptions.integrator_type`` / ``AcadosSimOpts.integrator_type`` + ``collocation_type`` -> ``IHM2MPC_INTEG_*``."""
    if integrator_type == "ERK":
        return INTEG_RK4
    if integrator_type == "IRK":
        if collocation_type == "GAUSS_LEGENDRE":
            return INTEG_IRK_GL4
        if collocation_type == "GAUSS_RADAU_IIA":
            return INTEG_IRK_RADAU4
        raise ValueError(f"collocation_type {collocation_type!r}: GAUSS_LEGENDRE or GAUSS_RADAU_IIA")
    raise ValueError(f"integrator_type {integrator_type!r}: ERK or IRK")


@dataclass
class ModelBounds:
    """``python/mpc.py:18-26`` (a pydantic model there)."""

    n_max: float = 2.0
    v_x_min: float = 0.0
    v_x_max: float = 31.0
    T_max: float = 500.0
    delta_max: float = 0.5
    T_dot_max: float = 1e6
    delta_dot_max: float = 1.0
    a_lat_max: float = 5.0


# ---- model wrappers (python/models.py:232-307, 455-606, 809-843) -------------------------------
def fkin6_model(x=None, u=None, p=None):
    """Frenet kinematic 6-DOF bicycle model; the arithmetic lives in the HIP kernels
    (``csrc/ihm2mpc_kernels.hip``).  Kept as a callable token so call sites read like the reference."""
    return "fkin6"


def fdyn6_model(xdot=None, x=None, u=None, p=None):
    """Frenet dynamic 4-wheel Pacejka model (implicit in the reference; solved for xdot on device)."""
    return "fdyn6"


def fdyn6u_model(xdot=None, x=None, u=None, p=None):
    """``fdyn6_model`` with every wheel's lateral force on its own slip angle.  The reference crosses them
    (``python/models.py:543-546``), which makes the model open-loop unstable; this variant is a named deviation."""
    return "fdyn6u"


def fdyn10_model(xdot=None, x=None, u=None, p=None):
    """Frenet model with wheel speeds: 15 states, 5 inputs (``python/models.py:609-801``) -- a PLANT of the MiL loop
    (``python/main.py:490-502``), never an OCP model."""
    return "fdyn10"


@dataclass
class AcadosModel:
    name: str = "ihm2_fkin6"
    kind: str = "fkin6"
    nx: int = NX
    nu: int = NU
    np_: int = 3000
    # nonlinear path constraints h(x) of old/generate_acaods_interface.py:191-212: None, or "track" for the two
    # track-boundary rows (right, left) at every stage and at the terminal stage (con_h_expr / con_h_expr_e there), or
    # "track+a_lat" for the kinematic model's set (:198-209): the same two rows plus the lateral acceleration a_lat at the
    # stages (lh, uh then have three entries: right, left, a_lat; lh_e, uh_e keep two).  The reference's T_dot and
    # delta_dot rows of that set are linear in (x, u): they are the general rows C, D, lg, ug here (python/mpc.py:92-99)
    con_h_expr: str | None = None
    car_length: float = car_length
    car_width: float = car_width

    @property
    def model_id(self) -> int:
        return _MODEL_IDS[self.kind]


def _dim(v, default):
    if v is None:
        return default
    if isinstance(v, int):
        return v
    return int(np.asarray(v).shape[0])


def get_acados_model_from_explicit_dynamics(name, continuous_model_fn, x=None, u=None, p=None) -> AcadosModel:
    """``python/models.py:809-825``.  ``x``, ``u``, ``p`` may be ints or arrays (only sizes matter)."""
    kind = continuous_model_fn() if callable(continuous_model_fn) else str(continuous_model_fn)
    if kind not in _MODEL_IDS:
        raise ValueError(f"unknown model {kind!r}")
    return AcadosModel(name=name, kind=kind, nx=_dim(x, NX), nu=_dim(u, NU), np_=_dim(p, 3000))


def get_acados_model_from_implicit_dynamics(name, continuous_model_fn, x=None, u=None, p=None) -> AcadosModel:
    """``python/models.py:828-843``."""
    return get_acados_model_from_explicit_dynamics(name, continuous_model_fn, x, u, p)


# ---- AcadosOcp-shaped containers -----------------------------------------------------------------
@dataclass
class AcadosOcpDims:
    N: int = 40
    nx: int = NX
    nu: int = NU
    np: int = 3000
    ny: int = NY
    ny_e: int = NX


@dataclass
class AcadosOcpCost:
    cost_type: str = "LINEAR_LS"
    cost_type_e: str = "LINEAR_LS"
    W: np.ndarray = field(default_factory=lambda: np.eye(NY))
    Vx: np.ndarray = field(default_factory=lambda: np.zeros((NY, NX)))
    Vu: np.ndarray = field(default_factory=lambda: np.zeros((NY, NU)))
    W_e: np.ndarray = field(default_factory=lambda: np.eye(NX))
    Vx_e: np.ndarray = field(default_factory=lambda: np.eye(NX))
    yref: np.ndarray = field(default_factory=lambda: np.ones(NY))
    yref_e: np.ndarray = field(default_factory=lambda: np.ones(NX))
    # slack penalties, one entry per soft constraint in the order [sbx..., sg..., sh...] (acados: zl s + 1/2 Zl s^2 ...);
    # terminal: [sbx_e..., sh_e...]
    zl: np.ndarray = field(default_factory=lambda: np.zeros(0))
    zu: np.ndarray = field(default_factory=lambda: np.zeros(0))
    Zl: np.ndarray = field(default_factory=lambda: np.zeros(0))
    Zu: np.ndarray = field(default_factory=lambda: np.zeros(0))
    zl_e: np.ndarray = field(default_factory=lambda: np.zeros(0))
    zu_e: np.ndarray = field(default_factory=lambda: np.zeros(0))
    Zl_e: np.ndarray = field(default_factory=lambda: np.zeros(0))
    Zu_e: np.ndarray = field(default_factory=lambda: np.zeros(0))


@dataclass
class AcadosOcpConstraints:
    x0: np.ndarray = field(default_factory=lambda: np.ones(NX))
    idxbx: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=int))
    lbx: np.ndarray = field(default_factory=lambda: np.zeros(0))
    ubx: np.ndarray = field(default_factory=lambda: np.zeros(0))
    idxbx_e: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=int))
    lbx_e: np.ndarray = field(default_factory=lambda: np.zeros(0))
    ubx_e: np.ndarray = field(default_factory=lambda: np.zeros(0))
    idxbu: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=int))
    lbu: np.ndarray = field(default_factory=lambda: np.zeros(0))
    ubu: np.ndarray = field(default_factory=lambda: np.zeros(0))
    C: np.ndarray = field(default_factory=lambda: np.zeros((NG, NX)))
    D: np.ndarray = field(default_factory=lambda: np.zeros((NG, NU)))
    lg: np.ndarray = field(default_factory=lambda: np.full(NG, -INF))
    ug: np.ndarray = field(default_factory=lambda: np.full(NG, INF))
    # soft constraints (acados names; old/generate_acaods_interface.py:380-449): positions within idxbx / the rows of C
    idxsbx: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=int))
    idxsbx_e: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=int))
    idxsg: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=int))
    # nonlinear rows lh <= h(x) <= uh (model.con_h_expr); the reference writes -1e3 for "no lower bound"
    lh: np.ndarray = field(default_factory=lambda: np.zeros(0))
    uh: np.ndarray = field(default_factory=lambda: np.zeros(0))
    lh_e: np.ndarray = field(default_factory=lambda: np.zeros(0))
    uh_e: np.ndarray = field(default_factory=lambda: np.zeros(0))
    idxsh: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=int))
    idxsh_e: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=int))


@dataclass
class AcadosOcpOptions:
    """Subset of ``acados_template.AcadosOcpOptions`` the reference sets (``python/main.py:227-238``,
    ``old/generate.py:18-26``) plus the knobs of this implementation."""

    tf: float = 2.0
    qp_solver: str = "PARTIAL_CONDENSING_HPIPM"   # accepted for drop-in; the QP is solved full-space (Riccati IPM)
    nlp_solver_type: str = "SQP_RTI"              # "SQP_RTI" (old/generate.py:21) or "SQP" (python/main.py:230)
    nlp_solver_max_iter: int = 1
    hessian_approx: str = "GAUSS_NEWTON"
    hpipm_mode: str = "SPEED_ABS"
    integrator_type: str = "ERK"                  # "ERK": RK4 x sim_method_num_steps (old/generate.py:23-25); "IRK": 4-stage collocation, one step
                                                  # per interval, 3 Newton iterations (python/main.py:234-236)
    collocation_type: str = "GAUSS_LEGENDRE"      # IRK only (acados' default); or "GAUSS_RADAU_IIA"
    sim_integrator_type: str = "ERK"              # integrator of the plant steps behind this solver (python/main.py:395-400: "IRK")
    sim_collocation_type: str = "GAUSS_RADAU_IIA"
    sim_method_num_stages: int = 4
    sim_method_num_steps: int = 25                # M; 1 is unstable on this model
    globalization: str = "FIXED_STEP"             # or "MERIT_BACKTRACKING" (python/main.py:237); SQP mode only
    alpha_min: float = 0.05                       # line search (acados' defaults)
    alpha_reduction: float = 0.7
    eps_sufficient_descent: float = 1e-4
    line_search_use_sufficient_descent: int = 0
    full_step_dual: int = 0
    nlp_solver_tol_stat: float | None = None      # SQP mode: None -> nlp_tol
    nlp_solver_tol_eq: float | None = None
    nlp_solver_tol_ineq: float | None = None
    nlp_solver_tol_comp: float | None = None
    print_level: int = 0
    # implementation knobs
    qp_solver_iter_max: int = 30
    qp_tol: float = 1e-6                          # relative (see csrc: scaled by |g|_inf and |b|_inf); ~sqrt(eps) is the fp64 limit
    qp_mu0: float = 0.1                # initial barrier parameter = qp_mu0 * max(1, |g|_inf)
    qp_tau0: float = 1.0               # lower clamp of the initial slacks (1/4 of the width for narrow two-sided constraints)
    cost_scale_stage: float | None = None         # None -> time step (acados scales stage cost by dt)
    nlp_tol: float = 1e-6                         # SQP mode: stop when all four KKT residuals <= tol


@dataclass
class AcadosOcp:
    model: AcadosModel = field(default_factory=AcadosModel)
    dims: AcadosOcpDims = field(default_factory=AcadosOcpDims)
    cost: AcadosOcpCost = field(default_factory=AcadosOcpCost)
    constraints: AcadosOcpConstraints = field(default_factory=AcadosOcpConstraints)
    solver_options: AcadosOcpOptions = field(default_factory=AcadosOcpOptions)
    parameter_values: np.ndarray | None = None
    code_export_directory: str = ""

    def flatten(self) -> "OcpData":
        return OcpData.from_ocp(self)


def get_acados_ocp(
    model: AcadosModel,
    Nf: int,
    n_max: float,
    v_x_max: float,
    T_max: float,
    delta_max: float,
    T_dot_max: float,
    delta_dot_max: float,
) -> AcadosOcp:
    """Same signature and content as ``python/mpc.py:29-101``."""
    ocp = AcadosOcp()
    ocp.model = model
    ocp.dims.N = Nf
    ocp.dims.nx = nx = model.nx
    ocp.dims.nu = nu = model.nu
    ocp.dims.np = model.np_
    # stage cost y = Vx x + Vu u = [x; u; x[-nu:] - u]  (mpc.py:49-58)
    ocp.dims.ny = ny = nx + nu + nu
    ocp.cost.W = np.eye(ny)
    Vx = np.zeros((ny, nx))
    Vx[:nx] = np.eye(nx)
    Vx[-nu:, -nu:] = np.eye(nu)
    Vu = np.zeros((ny, nu))
    Vu[-2 * nu:-nu] = np.eye(nu)
    Vu[-nu:] = -np.eye(nu)
    ocp.cost.Vx, ocp.cost.Vu = Vx, Vu
    ocp.dims.ny_e = nx
    ocp.cost.W_e = np.eye(nx)
    ocp.cost.Vx_e = np.eye(nx)
    ocp.cost.yref = np.ones(ny)
    ocp.cost.yref_e = np.ones(nx)
    ocp.constraints.x0 = np.ones(nx)
    ocp.parameter_values = np.ones(model.np_)
    # state boxes (mpc.py:79-84); the terminal index set [1,3,4,5] is reproduced as written (quirk Q1)
    c = ocp.constraints
    c.idxbx = np.array([1, 3, 6, 7])
    c.lbx = np.array([-n_max, 0.0, -T_max, -delta_max])
    c.ubx = np.array([n_max, v_x_max, T_max, delta_max])
    c.idxbx_e = np.array([1, 3, 4, 5])
    c.lbx_e = np.array([-n_max, -v_x_max, -T_max, -delta_max])
    c.ubx_e = np.array([n_max, v_x_max, T_max, delta_max])
    # control boxes (mpc.py:87-89)
    c.idxbu = np.array([0, 1])
    c.lbu = np.array([-T_max, -delta_max])
    c.ubu = np.array([T_max, delta_max])
    # rate rows g = C x + D u = u - x[6:8]  (mpc.py:92-99)
    c.C = np.zeros((2, nx))
    c.C[0, -2] = -1.0
    c.C[1, -1] = -1.0
    c.D = np.zeros((2, nu))
    c.D[0, 0] = 1.0
    c.D[1, 1] = 1.0
    c.lg = np.array([t_T * -T_dot_max, t_delta * -delta_dot_max])
    c.ug = np.array([t_T * T_dot_max, t_delta * delta_dot_max])
    return ocp


# ---- dense, per-stage arrays handed to the C-ABI ---------------------------------------------------
@dataclass
class OcpData:
    N: int
    M: int
    dt: float
    model: int
    integrator: int
    sim_integrator: int
    cost_scale_stage: float
    W: np.ndarray      # (N,12,12)
    W_e: np.ndarray    # (8,8)
    lbx: np.ndarray    # (N+1,8), +-inf = absent, row 0 unused
    ubx: np.ndarray
    lbu: np.ndarray    # (N,2)
    ubu: np.ndarray
    C: np.ndarray      # (N,2,8)
    D: np.ndarray      # (N,2,2)
    lg: np.ndarray     # (N,2)
    ug: np.ndarray
    ipm_iter_max: int
    ipm_tol: float
    ipm_mu0: float
    ipm_tau0: float
    nlp_solver_type: str = "SQP_RTI"
    nlp_solver_max_iter: int = 1
    nlp_tol: float = 1e-6
    globalization: str = "FIXED_STEP"
    alpha_min: float = 0.05
    alpha_reduction: float = 0.7
    eps_sufficient_descent: float = 1e-4
    use_sufficient_descent: int = 0
    full_step_dual: int = 0
    sqp_tol: np.ndarray = field(default_factory=lambda: np.full(4, 1e-6))   # stat, eq, ineq, comp
    soft_z: np.ndarray | None = None   # (N+1,28) linear slack penalty per one-sided constraint (14 lower, 14 upper)
    soft_Z: np.ndarray | None = None   # (N+1,28) quadratic slack penalty; < 0 = hard side
    path_on: int = 0                   # track-boundary rows h (stages 1..N)
    alat_on: int = 0                   # lateral-acceleration row (stages 1..N-1, kinematic model): a fifteenth row beside the 14 of the tables
    alat_lb: float = -INF
    alat_ub: float = INF
    alat_soft_z: np.ndarray | None = None   # (2) linear slack penalty of its lower / upper side
    alat_soft_Z: np.ndarray | None = None   # (2) quadratic one; < 0 = hard side; None = both hard
    car_L: float = car_length
    car_W: float = car_width
    lh: np.ndarray = field(default_factory=lambda: np.full(NH, -INF))
    uh: np.ndarray = field(default_factory=lambda: np.full(NH, INF))

    @staticmethod
    def from_ocp(ocp: AcadosOcp) -> "OcpData":
        o, d, c = ocp.solver_options, ocp.dims, ocp.constraints
        N = d.N
        if ocp.model.kind == "fdyn10":
            raise ValueError("fdyn10 is a plant model (python/main.py:490-502): it has no OCP in the reference and none here")
        if (d.nx, d.nu, d.ny, d.ny_e) != (NX, NU, NY, NX):
            raise ValueError("this implementation is specialised to nx=8, nu=2, ny=12, ny_e=8")
        if ocp.cost.cost_type != "LINEAR_LS" or ocp.cost.cost_type_e != "LINEAR_LS":
            raise ValueError("only LINEAR_LS costs (python/mpc.py:49,62)")
        ref = get_acados_ocp(ocp.model, N, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0)
        if not (np.array_equal(ocp.cost.Vx, ref.cost.Vx) and np.array_equal(ocp.cost.Vu, ref.cost.Vu)
                and np.array_equal(ocp.cost.Vx_e, np.eye(NX))):
            raise ValueError("output selectors Vx, Vu, Vx_e are fixed to those of python/mpc.py:51-64")
        integ = integrator_code(o.integrator_type, o.collocation_type)
        sim_integ = integrator_code(o.sim_integrator_type, o.sim_collocation_type)
        if o.sim_method_num_stages != 4:
            raise ValueError("4 stages: the classical RK4 (ERK) or 4-stage collocation (IRK), python/main.py:235")
        if integ != INTEG_RK4 and o.sim_method_num_steps != 1:
            raise ValueError("IRK takes one step per shooting interval (sim_method_num_steps = 1, python/main.py:236)")
        if o.nlp_solver_type not in ("SQP_RTI", "SQP"):
            raise ValueError(f"nlp_solver_type {o.nlp_solver_type!r}")
        if o.globalization not in ("FIXED_STEP", "MERIT_BACKTRACKING"):
            raise ValueError(f"globalization {o.globalization!r}")
        dt = o.tf / N
        lbx = np.full((N + 1, NX), -INF)
        ubx = np.full((N + 1, NX), INF)
        idx = np.asarray(c.idxbx, dtype=int)
        lbx[1:N, idx] = np.asarray(c.lbx, dtype=float)
        ubx[1:N, idx] = np.asarray(c.ubx, dtype=float)
        idxe = np.asarray(c.idxbx_e, dtype=int)
        lbx[N, idxe] = np.asarray(c.lbx_e, dtype=float)
        ubx[N, idxe] = np.asarray(c.ubx_e, dtype=float)
        lbu = np.full((N, NU), -INF)
        ubu = np.full((N, NU), INF)
        idu = np.asarray(c.idxbu, dtype=int)
        lbu[:, idu] = np.asarray(c.lbu, dtype=float)
        ubu[:, idu] = np.asarray(c.ubu, dtype=float)
        ng = np.asarray(c.C).shape[0]
        if ng > NG:
            raise ValueError(f"at most {NG} general constraint rows")
        Cm = np.zeros((NG, NX)); Dm = np.zeros((NG, NU)); lg = np.full(NG, -INF); ug = np.full(NG, INF)
        Cm[:ng] = c.C; Dm[:ng] = c.D; lg[:ng] = c.lg; ug[:ng] = c.ug
        # soft sides: slot rows 0..7 state boxes, 8..9 input boxes, 10..11 general rows
        soft_z = soft_Z = None
        cost = ocp.cost
        # nonlinear track rows: the same two rows at the stages and at the terminal stage
        path_on, lh, uh = 0, np.full(NH, -INF), np.full(NH, INF)
        alat_on, alat_lb, alat_ub, alat_soft_z, alat_soft_Z = 0, -INF, INF, None, None
        if ocp.model.con_h_expr is not None:
            if ocp.model.con_h_expr not in ("track", "track+a_lat"):
                raise ValueError("con_h_expr must be None, 'track' or 'track+a_lat'")
            alat_on = int(ocp.model.con_h_expr == "track+a_lat")
            lh, uh = np.asarray(c.lh, dtype=float), np.asarray(c.uh, dtype=float)
            if lh.shape != (NH + alat_on,) or uh.shape != (NH + alat_on,):
                raise ValueError(f"lh, uh need {NH + alat_on} entries (right, left track row" + (", a_lat)" if alat_on else ")"))
            if alat_on:
                if ocp.model.kind != "fkin6":
                    raise ValueError("a_lat is a row of the kinematic model only (old/generate_acaods_interface.py:206)")
                if o.nlp_solver_type != "SQP_RTI":
                    raise ValueError("the a_lat row is implemented for SQP_RTI (old/generate.py:21)")
                alat_lb = float(lh[2]) if abs(lh[2]) < 1e20 else -INF
                alat_ub = float(uh[2]) if abs(uh[2]) < 1e20 else INF
                lh, uh = lh[:NH], uh[:NH]
            if not (np.array_equal(lh, np.asarray(c.lh_e, dtype=float)) and np.array_equal(uh, np.asarray(c.uh_e, dtype=float))):
                raise ValueError("lh_e, uh_e must equal the track rows of lh, uh (old/generate_acaods_interface.py:411-449)")
            path_on = 1
        elif len(c.lh) or len(c.idxsh) or len(c.idxsh_e):
            raise ValueError("lh/uh/idxsh given but model.con_h_expr is None")
        nsbx, nsg, nsh, nsbx_e, nsh_e = len(c.idxsbx), len(c.idxsg), len(c.idxsh), len(c.idxsbx_e), len(c.idxsh_e)
        if nsbx + nsg + nsh + nsbx_e + nsh_e > 0:
            ns, ns_e = nsbx + nsg + nsh, nsbx_e + nsh_e
            soft_z = np.zeros((N + 1, NLAM)); soft_Z = np.full((N + 1, NLAM), -1.0)
            if len(cost.zl) != ns or len(cost.Zl) != ns or len(cost.zu) != ns or len(cost.Zu) != ns:
                raise ValueError("cost.zl/zu/Zl/Zu need one entry per soft constraint [sbx..., sg..., sh...]")
            if len(cost.zl_e) != ns_e or len(cost.Zl_e) != ns_e or len(cost.zu_e) != ns_e or len(cost.Zu_e) != ns_e:
                raise ValueError("cost.zl_e/zu_e/Zl_e/Zu_e need one entry per terminal soft constraint [sbx_e..., sh_e...]")

            def put(stages, row, j, e=""):
                soft_z[stages, row], soft_Z[stages, row] = getattr(cost, "zl" + e)[j], getattr(cost, "Zl" + e)[j]
                soft_z[stages, NC + row], soft_Z[stages, NC + row] = getattr(cost, "zu" + e)[j], getattr(cost, "Zu" + e)[j]

            for j, pos in enumerate(np.asarray(c.idxsbx, dtype=int)):
                put(slice(1, N), int(idx[pos]), j)
            for j, pos in enumerate(np.asarray(c.idxsg, dtype=int)):
                put(slice(0, N), 10 + int(pos), nsbx + j)
            for j, pos in enumerate(np.asarray(c.idxsh, dtype=int)):
                if int(pos) == NH and alat_on:      # the a_lat row keeps its penalties beside the 28 columns
                    if alat_soft_z is None:
                        alat_soft_z, alat_soft_Z = np.zeros(2), np.full(2, -1.0)
                    alat_soft_z[:] = cost.zl[nsbx + nsg + j], cost.zu[nsbx + nsg + j]
                    alat_soft_Z[:] = cost.Zl[nsbx + nsg + j], cost.Zu[nsbx + nsg + j]
                    continue
                if not 0 <= int(pos) < NH:
                    raise ValueError(f"idxsh position {int(pos)} names no row of con_h_expr")
                put(slice(1, N), 12 + int(pos), nsbx + nsg + j)
            for j, pos in enumerate(np.asarray(c.idxsbx_e, dtype=int)):
                put(N, int(idxe[pos]), j, "_e")
            for j, pos in enumerate(np.asarray(c.idxsh_e, dtype=int)):
                put(N, 12 + int(pos), nsbx_e + j, "_e")
        return OcpData(
            soft_z=soft_z, soft_Z=soft_Z, path_on=path_on, car_L=float(ocp.model.car_length), car_W=float(ocp.model.car_width),
            lh=lh, uh=uh, alat_on=alat_on, alat_lb=alat_lb, alat_ub=alat_ub, alat_soft_z=alat_soft_z, alat_soft_Z=alat_soft_Z,
            N=N, M=int(o.sim_method_num_steps), dt=dt, model=ocp.model.model_id,
            integrator=integ, sim_integrator=sim_integ,
            cost_scale_stage=dt if o.cost_scale_stage is None else float(o.cost_scale_stage),
            W=np.tile(np.asarray(ocp.cost.W, dtype=float)[None], (N, 1, 1)), W_e=np.array(ocp.cost.W_e, dtype=float),
            lbx=lbx, ubx=ubx, lbu=lbu, ubu=ubu,
            C=np.tile(Cm[None], (N, 1, 1)), D=np.tile(Dm[None], (N, 1, 1)),
            lg=np.tile(lg[None], (N, 1)), ug=np.tile(ug[None], (N, 1)),
            ipm_iter_max=int(o.qp_solver_iter_max), ipm_tol=float(o.qp_tol), ipm_mu0=float(o.qp_mu0),
            ipm_tau0=float(o.qp_tau0), nlp_solver_type=o.nlp_solver_type,
            nlp_solver_max_iter=int(o.nlp_solver_max_iter), nlp_tol=float(o.nlp_tol),
            globalization=o.globalization, alpha_min=float(o.alpha_min), alpha_reduction=float(o.alpha_reduction),
            eps_sufficient_descent=float(o.eps_sufficient_descent), use_sufficient_descent=int(o.line_search_use_sufficient_descent),
            full_step_dual=int(o.full_step_dual),
            sqp_tol=np.array([float(o.nlp_tol) if t is None else float(t) for t in
                              (o.nlp_solver_tol_stat, o.nlp_solver_tol_eq, o.nlp_solver_tol_ineq, o.nlp_solver_tol_comp)]),
        )

    def as_dict(self, s_ref, kappa_ref, track_widths=None) -> dict:
        """Plain description (arrays + scalars) incl. the track tables (used by the test suite to describe the same problem to its CPU checker)."""
        d = {k: getattr(self, k) for k in (
            "N", "M", "dt", "model", "integrator", "cost_scale_stage", "W", "W_e", "lbx", "ubx", "lbu", "ubu",
            "C", "D", "lg", "ug", "ipm_iter_max", "ipm_tol", "ipm_mu0", "ipm_tau0")}
        d["soft_z"], d["soft_Z"] = self.soft_z, self.soft_Z
        d["s_ref"] = np.atleast_2d(np.asarray(s_ref, dtype=float))
        d["kappa_ref"] = np.atleast_2d(np.asarray(kappa_ref, dtype=float))
        d["path_on"], d["car_L"], d["car_W"], d["lh"], d["uh"] = self.path_on, self.car_L, self.car_W, self.lh, self.uh
        d["alat_on"], d["alat_lb"], d["alat_ub"] = self.alat_on, self.alat_lb, self.alat_ub
        d["alat_soft_z"], d["alat_soft_Z"] = self.alat_soft_z, self.alat_soft_Z
        if self.path_on:
            if track_widths is None:
                raise ValueError("track rows need track_widths (ntracks, 2) = (right, left)")
            d["widths"] = np.atleast_2d(np.asarray(track_widths, dtype=float))
        return d


def default_weights(
    q_s=1.0, q_n=1.0, q_psi=1.0, q_v_x=1.0, q_v_y=1.0, q_r=1.0, q_T=1.0, q_delta=100.0,
    q_s_f=1000.0, q_n_f=100.0, q_psi_f=100.0, q_v_x_f=1.0, q_v_y_f=1.0, q_r_f=1.0, q_T_f=1.0, q_delta_f=100.0,
    q_T_dot=0.0, q_delta_dot=500.0,
):
    """Stage and terminal weight matrices of ``IHM2Controller.__init__`` (``python/main.py:193-210,255-295``)."""
    W = np.diag([q_s, q_n, q_psi, q_v_x, q_v_y, q_r, q_T, q_delta, q_T, q_delta, q_T_dot, q_delta_dot])
    W_e = np.diag([q_s_f, q_n_f, q_psi_f, q_v_x_f, q_v_y_f, q_r_f, q_T_f, q_delta_f])
    return W, W_e
