"""Parameter and state records of the reference's ``new_python`` rewrite (``new_python/controller.py:11-131``: ``@PODS`` classes;
``strongpods`` is not installed here, plain dataclasses carry the same field names and helper properties), filled from this build's
constants (``constants.py`` = ``python/constants.py:43-111``).  ``CarState`` is the 15-entry Cartesian state of the 10-DOF models
(six rigid-body states, four wheel speeds, four wheel torques, steering); ``to_frenet_dyn10`` / ``from_array`` connect it to the
``fdyn10`` plant of ``ihm2mpc_sim_step_dyn10`` (state order of ``python/models.py:613-627``: wheels FL, FR, RL, RR)."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import constants as K


@dataclass
class PacejkaCoefficients:
    b1: float = 0.0
    b2: float = 0.0
    b3: float = 0.0
    c1: float = 0.0
    d1: float = 0.0
    d2: float = 0.0
    e1: float = 0.0
    e2: float = 0.0
    e3: float = 0.0
    B: float = 0.0
    C: float = 0.0
    D: float = 0.0
    BCD: float = 0.0

    @property
    def cornering_stiffness(self) -> float:
        return self.BCD


@dataclass
class TireParams:
    radius: float
    inertia: float
    longitudinal_pacejka_coefficients: PacejkaCoefficients
    lateral_pacejka_coefficients: PacejkaCoefficients


@dataclass
class CarGeometry:
    length: float
    width: float
    wheelbase: float
    track: float
    cog_to_rear_axle: float
    cog_to_front_axle: float
    cog_height: float


@dataclass
class DrivetrainParams:
    C_m0: float
    C_r0: float
    C_r1: float
    C_r2: float


@dataclass
class AerodynamicParams:
    downforce_coeff: float


@dataclass
class ActuatorParams:
    wheel_torque_max: float
    total_torque_max: float
    steering_max: float
    steering_rate_max: float
    steering_time_constant: float


@dataclass
class CarParams:
    mass: float
    yaw_inertia: float
    geometry: CarGeometry
    tire_params: TireParams
    drivetrain_params: DrivetrainParams
    aerodynamic_params: AerodynamicParams
    actuator_params: ActuatorParams


def default_car_params() -> CarParams:
    """The ihm2 car (``python/constants.py:43-111``, bounds of ``python/main.py:55-64``)."""
    lon = PacejkaCoefficients(b1=K.b1s, b2=K.b2s, b3=K.b3s, c1=K.c1s, d1=K.d1s, d2=K.d2s, e1=K.e1s, e2=K.e2s, e3=K.e3s, B=K.Bs, C=K.Cs, D=K.Ds, BCD=K.BCDs)
    lat = PacejkaCoefficients(b1=K.b1a, b2=K.b2a, c1=K.c1a, d1=K.d1a, d2=K.d2a, e1=K.e1a, e2=K.e2a, B=K.Ba, C=K.Ca, D=K.Da, BCD=K.BCDa)
    return CarParams(
        mass=K.m, yaw_inertia=K.I_z,
        geometry=CarGeometry(K.car_length, K.car_width, K.wheelbase, K.axle_track, K.l_R, K.l_F, K.z_CG),
        tire_params=TireParams(K.R_w, K.I_w, lon, lat),
        drivetrain_params=DrivetrainParams(K.C_m0, K.C_r0, K.C_r1, K.C_r2),
        aerodynamic_params=AerodynamicParams(K.C_downforce),
        actuator_params=ActuatorParams(wheel_torque_max=125.0, total_torque_max=500.0, steering_max=0.5, steering_rate_max=1.0, steering_time_constant=K.t_delta),
    )


_CAR_STATE_FIELDS = ("X", "Y", "phi", "v_x", "v_y", "r", "omega_FR", "omega_FL", "omega_RR", "omega_RL", "tau_FR", "tau_FL", "tau_RR", "tau_RL", "delta")


@dataclass
class CarState:
    X: float = 0.0
    Y: float = 0.0
    phi: float = 0.0
    v_x: float = 0.0
    v_y: float = 0.0
    r: float = 0.0
    omega_FR: float = 0.0
    omega_FL: float = 0.0
    omega_RR: float = 0.0
    omega_RL: float = 0.0
    tau_FR: float = 0.0
    tau_FL: float = 0.0
    tau_RR: float = 0.0
    tau_RL: float = 0.0
    delta: float = 0.0

    @property
    def position(self) -> np.ndarray:
        return np.array([self.X, self.Y])

    @property
    def pose(self) -> np.ndarray:
        return np.array([self.X, self.Y, self.phi])

    @property
    def v(self) -> float:
        return float(np.hypot(self.v_x, self.v_y))

    @property
    def T(self) -> float:
        return self.tau_FR + self.tau_FL + self.tau_RR + self.tau_RL

    def to_array(self) -> np.ndarray:
        return np.array([getattr(self, f) for f in _CAR_STATE_FIELDS], dtype=np.float64)

    @classmethod
    def from_array(cls, a) -> "CarState":
        a = np.asarray(a, dtype=np.float64).reshape(-1)
        if a.size != len(_CAR_STATE_FIELDS):
            raise ValueError(f"{len(_CAR_STATE_FIELDS)} entries expected")
        return cls(**{f: float(v) for f, v in zip(_CAR_STATE_FIELDS, a)})

    def to_frenet_dyn10(self, s: float, n: float, psi: float) -> np.ndarray:
        """The ``fdyn10`` state (``python/models.py:613-627``) of this car at the Frenet pose ``(s, n, psi)``: wheels in the order FL, FR, RL, RR."""
        return np.array([s, n, psi, self.v_x, self.v_y, self.r, self.omega_FL, self.omega_FR, self.omega_RL, self.omega_RR,
                         self.tau_FL, self.tau_FR, self.tau_RL, self.tau_RR, self.delta], dtype=np.float64)


@dataclass
class MotionPlanPoints:
    """``new_python/controller.py:119-126`` (``MotionPlan`` there; this build's ``track.MotionPlan`` is the track table of ``python/``)."""

    X: np.ndarray = field(default_factory=lambda: np.zeros(0))
    Y: np.ndarray = field(default_factory=lambda: np.zeros(0))
    phi: np.ndarray = field(default_factory=lambda: np.zeros(0))
    v_x: np.ndarray = field(default_factory=lambda: np.zeros(0))
    v_y: np.ndarray = field(default_factory=lambda: np.zeros(0))
    r: np.ndarray = field(default_factory=lambda: np.zeros(0))


@dataclass
class TrackingPlan:
    pass
