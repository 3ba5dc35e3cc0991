"""CPU ORACLE helper (test infrastructure, NOT the product).

NumPy mirror of the reference's vehicle ODEs, written a second time, independently of the C oracle,
directly from ``python/models.py:232-307`` (fkin6) and ``:455-606`` (fdyn6, here left in its
IMPLICIT residual form ``F(xdot, x, u) = 0`` exactly as the reference writes it).  Works on real and
complex inputs so tests can take complex-step derivatives.  Used only by tests to cross-check the C
oracle: (a) ``f``, (b) its Jacobian, (c) that the oracle's explicit fdyn6 solves the reference's
implicit residual.
"""
import numpy as np

# python/constants.py:43-111
g = 9.81
m = 230.0
I_z = 137.583
z_CG = 0.295
axle_track = front_axle_track = rear_axle_track = 1.24
l_R = 0.7853
l_F = 0.7853
wheelbase = 1.5706
rear_weight_distribution = l_R / wheelbase
C_m0, C_r0, C_r1, C_r2 = 4.950, 297.030, 16.665, 0.6784
b1a, b2a, c1a, d1a, d2a, e1a, e2a = 3.79e1, 5.28e2, 1.57, -2.03e-4, 1.77, -2.24e-3, 1.81
static_weight = 0.5 * m * g * l_F / wheelbase
BCDa = b1a * np.sin(2 * np.arctan(static_weight / b2a))
Ca = c1a
Da = d1a * static_weight + d2a
Ea = e1a * static_weight + e2a
Ba = BCDa / (Ca * Da)
t_T = 1e-3
t_delta = 0.02
C_downforce = 3.96864
K_tv = 300.0


def smooth_sgn(x):
    return np.tanh(10.0 * x)


def smooth_abs_nonzero(x, min_val=1e-6):
    return smooth_sgn(x) * x + min_val * np.exp(-x * x)


def lat_pacejka(alpha):
    return Da * np.sin(Ca * np.arctan(Ba * alpha - Ea * (Ba * alpha - np.arctan(Ba * alpha))))


def kappa_interp(s, s_ref, kappa_ref):
    """casadi 'linear' interpolant, exact lookup, linear extrapolation; complex-safe in s."""
    sr = np.real(s)
    n = len(s_ref)
    i = int(np.clip(np.searchsorted(s_ref, sr, side="right") - 1, 0, n - 2))
    slope = (kappa_ref[i + 1] - kappa_ref[i]) / (s_ref[i + 1] - s_ref[i])
    return kappa_ref[i] + slope * (s - s_ref[i])


def catan2_pos(y, x):
    """atan2 for Re(x) > 0, analytic in both arguments."""
    return np.arctan(y / x)


def fkin6(x, u, s_ref, kappa_ref):
    s, n, psi, v_x, v_y, r, T, delta = x
    u_T, u_delta = u
    delta_dot = (u_delta - delta) / t_delta
    T_dot = (u_T - T) / t_T
    F_motor = C_m0 * T
    F_drag = -(C_r0 + C_r1 * v_x + C_r2 * v_x * v_x) * smooth_sgn(v_x)
    F_Rx = 0.5 * F_motor + F_drag
    F_Fx = 0.5 * F_motor
    tandelta = np.tan(delta)
    beta = np.arctan(rear_weight_distribution * tandelta)
    cos_beta, sin_beta = np.cos(beta), np.sin(beta)
    beta_dot = (rear_weight_distribution * (1 + tandelta * tandelta)
                / (1 + rear_weight_distribution * rear_weight_distribution * tandelta * tandelta) * delta_dot)
    v_dot = (F_Rx * cos_beta + F_Fx * np.cos(delta - beta)) / m
    kap = kappa_interp(s, s_ref, kappa_ref)
    s_dot = (v_x * np.cos(psi) - v_y * np.sin(psi)) / (1 + kap * n)
    v_y_dot = v_dot * sin_beta + beta_dot * v_x
    return np.array([
        s_dot,
        v_x * np.sin(psi) + v_y * np.cos(psi),
        r - kap * s_dot,
        v_dot * cos_beta - beta_dot * v_y,
        v_y_dot,
        l_R * v_y_dot - beta_dot,
        T_dot,
        delta_dot,
    ])


def kin6(x, u, s_ref=None, kappa_ref=None):
    """Cartesian kinematic model, python/models.py:168-229."""
    X, Y, phi, v_x, v_y, r, T, delta = x
    u_T, u_delta = u
    delta_dot = (u_delta - delta) / t_delta
    T_dot = (u_T - T) / t_T
    F_motor = C_m0 * T
    F_drag = -(C_r0 + C_r1 * v_x + C_r2 * v_x * v_x) * smooth_sgn(v_x)
    F_Rx = 0.5 * F_motor + F_drag
    F_Fx = 0.5 * F_motor
    tandelta = np.tan(delta)
    rw = rear_weight_distribution
    beta = np.arctan(rw * tandelta)
    sinbeta = rw * tandelta / np.sqrt(1 + rw * rw * tandelta * tandelta)
    cosbeta = 1 / np.sqrt(1 + rw * rw * tandelta * tandelta)
    beta_dot = rw * (1 + tandelta * tandelta) / (1 + rw * rw * tandelta * tandelta) * delta_dot
    v_dot = (F_Rx * cosbeta + F_Fx * np.cos(delta - beta)) / m
    return np.array([
        v_x * np.cos(phi) - v_y * np.sin(phi),
        v_x * np.sin(phi) + v_y * np.cos(phi),
        r,
        v_dot * cosbeta - beta_dot * v_y,
        v_dot * sinbeta + beta_dot * v_x,
        (v_dot * sinbeta + beta_dot * v_x) / l_R,
        T_dot,
        delta_dot,
    ])


def dyn6_residual(xdot, x, u):
    """Implicit residual of the Cartesian dynamic model in the order of python/models.py:421-452."""
    X, Y, phi, v_x, v_y, r, T, delta = x
    u_T, u_delta = u
    X_dot, Y_dot, phi_dot, v_x_dot, v_y_dot, r_dot, T_dot, delta_dot = xdot
    a_x = v_x_dot - v_y * r
    a_y = v_y_dot + v_x * r
    F_downforce = 0.5 * C_downforce * v_x * v_x
    lon_wt = 0.5 * m * a_x * z_CG / wheelbase
    lat_wt = 0.5 * m * a_y * z_CG / axle_track
    F_z_FL = -(static_weight - lon_wt + lat_wt + 0.25 * F_downforce)
    F_z_FR = -(static_weight - lon_wt - lat_wt + 0.25 * F_downforce)
    F_z_RL = -(static_weight + lon_wt + lat_wt + 0.25 * F_downforce)
    F_z_RR = -(static_weight + lon_wt - lat_wt + 0.25 * F_downforce)
    v_x_FL = v_x - 0.5 * axle_track * r
    v_x_FR = v_x + 0.5 * axle_track * r
    v_x_RL, v_x_RR = v_x_FL, v_x_FR
    v_y_FL = v_y_FR = v_y + l_F * r
    v_y_RL = v_y_RR = v_y - l_R * r
    alpha_FL = np.arctan2(v_y_FL, v_x_FL) - delta
    alpha_FR = np.arctan2(v_y_FR, v_x_FR) - delta
    alpha_RL = np.arctan2(v_y_RL, v_x_RL)
    alpha_RR = np.arctan2(v_y_RR, v_x_RR)
    F_y_FL, F_y_FR = F_z_FL * lat_pacejka(alpha_FL), F_z_FR * lat_pacejka(alpha_FR)
    F_y_RL, F_y_RR = F_z_RL * lat_pacejka(alpha_RL), F_z_RR * lat_pacejka(alpha_RR)
    F_drag = -(C_r0 + C_r1 * v_x + C_r2 * v_x * v_x) * np.tanh(1000 * v_x)
    tandelta = np.tan(delta)
    rw = rear_weight_distribution
    sinbeta = rw * tandelta / np.sqrt(1 + rw * rw * tandelta * tandelta)
    delta_tau = K_tv * (v_x * sinbeta / l_R - r)
    den = -m * g - 0.25 * F_downforce
    F_x_FL = C_m0 * (T - delta_tau) * F_z_FL / den
    F_x_FR = C_m0 * (T + delta_tau) * F_z_FR / den
    F_x_RL = C_m0 * (T - delta_tau) * F_z_RL / den
    F_x_RR = C_m0 * (T + delta_tau) * F_z_RR / den
    cd, sd = np.cos(delta), np.sin(delta)
    return np.array([
        X_dot - (v_x * np.cos(phi) - v_y * np.sin(phi)),
        Y_dot - (v_x * np.sin(phi) + v_y * np.cos(phi)),
        phi_dot - r,
        m * a_x - ((F_x_FR + F_x_FL) * cd - (F_y_FR + F_y_FL) * sd + F_x_RR + F_x_RL + F_drag),
        m * a_y - ((F_x_FR + F_x_FL) * sd + (F_y_FR + F_y_FL) * cd + F_y_RR + F_y_RL),
        I_z * r_dot - (
            (F_x_FR * cd - F_y_FR * sd) * axle_track / 2 + (F_x_FR * sd + F_y_FR * cd) * l_F
            - (F_x_FL * cd - F_y_FL * sd) * axle_track / 2 + (F_x_FL * sd + F_y_FL * cd) * l_F
            + F_x_RR * axle_track / 2 - F_y_RR * l_R - F_x_RL * axle_track / 2 - F_y_RL * l_R
        ),
        T_dot - (u_T - T) / t_T,
        delta_dot - (u_delta - delta) / t_delta,
    ])


def fdyn6_residual(xdot, x, u, s_ref, kappa_ref, uncrossed=False):
    """Implicit residual exactly in the order of python/models.py:574-606."""
    s, n, psi, v_x, v_y, r, T, delta = x
    u_T, u_delta = u
    s_dot, n_dot, psi_dot, v_x_dot, v_y_dot, r_dot, T_dot, delta_dot = xdot
    a_x = v_x_dot - v_y * r
    a_y = v_y_dot + v_x * r
    F_downforce = 0.5 * C_downforce * v_x * v_x
    lon_wt = 0.5 * m * a_x * z_CG / wheelbase
    lat_wt = 0.5 * m * a_y * z_CG / axle_track
    F_z_FL = -(static_weight - lon_wt + lat_wt + 0.25 * F_downforce)
    F_z_FR = -(static_weight - lon_wt - lat_wt + 0.25 * F_downforce)
    F_z_RL = -(static_weight + lon_wt + lat_wt + 0.25 * F_downforce)
    F_z_RR = -(static_weight + lon_wt - lat_wt + 0.25 * F_downforce)
    v_x_FL = v_x - 0.5 * front_axle_track * r
    v_x_FR = v_x + 0.5 * front_axle_track * r
    v_y_FL = v_y + l_F * r
    v_y_FR = v_y + l_F * r
    cd, sd = np.cos(delta), np.sin(delta)
    v_lon_FL = cd * v_x_FL + sd * v_y_FL
    v_lon_FR = cd * v_x_FR + sd * v_y_FR
    v_lat_FL = -sd * v_x_FL + cd * v_y_FL
    v_lat_FR = -sd * v_x_FR + cd * v_y_FR
    v_lon_RL = v_x - 0.5 * rear_axle_track * r
    v_lon_RR = v_x + 0.5 * rear_axle_track * r
    v_lat_RL = v_y - l_R * r
    v_lat_RR = v_y - l_R * r
    alpha_FL = catan2_pos(v_lat_FL, smooth_abs_nonzero(v_lon_FL))
    alpha_FR = catan2_pos(v_lat_FR, smooth_abs_nonzero(v_lon_FR))
    alpha_RL = catan2_pos(v_lat_RL, smooth_abs_nonzero(v_lon_RL))
    alpha_RR = catan2_pos(v_lat_RR, smooth_abs_nonzero(v_lon_RR))
    F_lat_FL = F_z_FL * lat_pacejka(alpha_RR)
    F_lat_FR = F_z_FR * lat_pacejka(alpha_RL)
    F_lat_RL = F_z_RL * lat_pacejka(alpha_FR)
    F_lat_RR = F_z_RR * lat_pacejka(alpha_FL)
    if uncrossed:       # model "fdyn6u": every wheel with its own slip angle (named deviation from quirk Q3)
        F_lat_FL, F_lat_FR = F_z_FL * lat_pacejka(alpha_FL), F_z_FR * lat_pacejka(alpha_FR)
        F_lat_RL, F_lat_RR = F_z_RL * lat_pacejka(alpha_RL), F_z_RR * lat_pacejka(alpha_RR)
    F_drag = -(C_r0 + C_r1 * v_x + C_r2 * v_x * v_x) * smooth_sgn(v_x)
    beta = np.arctan(rear_weight_distribution * np.tan(delta))
    r_kin = np.sqrt(v_x * v_x + v_y * v_y) * np.sin(beta) / l_R
    delta_tau = K_tv * (r_kin - r)
    den = -m * g - 0.25 * F_downforce
    F_lon_FL = C_m0 * (T - delta_tau) * F_z_FL / den
    F_lon_FR = C_m0 * (T + delta_tau) * F_z_FR / den
    F_lon_RL = C_m0 * (T - delta_tau) * F_z_RL / den
    F_lon_RR = C_m0 * (T + delta_tau) * F_z_RR / den
    kap = kappa_interp(s, s_ref, kappa_ref)
    s_dot_expr = (v_x * np.cos(psi) - v_y * np.sin(psi)) / (1 + kap * n)
    return np.array([
        s_dot - s_dot_expr,
        n_dot - (v_x * np.sin(psi) + v_y * np.cos(psi)),
        psi_dot - (r - kap * s_dot_expr),
        m * a_x - ((F_lon_FR + F_lon_FL) * cd - (F_lat_FR + F_lat_FL) * sd + F_lon_RR + F_lon_RL + F_drag),
        m * a_y - ((F_lon_FR + F_lon_FL) * sd + (F_lat_FR + F_lat_FL) * cd + F_lat_RR + F_lat_RL),
        I_z * r_dot - (
            (F_lon_FR * cd - F_lat_FR * sd) * axle_track / 2
            + (F_lon_FR * sd + F_lat_FR * cd) * l_F
            - (F_lon_FL * cd - F_lat_FL * sd) * axle_track / 2
            + (F_lon_FL * sd + F_lat_FL * cd) * l_F
            + F_lon_RR * axle_track / 2
            - F_lat_RR * l_R
            - F_lon_RL * axle_track / 2
            - F_lat_RL * l_R
        ),
        T_dot - (u_T - T) / t_T,
        delta_dot - (u_delta - delta) / t_delta,
    ])


def fdyn6(x, u, s_ref, kappa_ref, uncrossed=False):
    """Explicit xdot: the residual is affine in xdot, so one linear solve is exact."""
    x = np.asarray(x)
    dtype = np.result_type(x.dtype, np.asarray(u).dtype, np.float64)
    r0 = fdyn6_residual(np.zeros(8, dtype=dtype), x, u, s_ref, kappa_ref, uncrossed)
    Jm = np.zeros((8, 8), dtype=dtype)
    for j in range(8):
        e = np.zeros(8, dtype=dtype)
        e[j] = 1.0
        Jm[:, j] = fdyn6_residual(e, x, u, s_ref, kappa_ref, uncrossed) - r0
    return np.linalg.solve(Jm, -r0)


def fdyn6u(x, u, s_ref, kappa_ref):
    return fdyn6(x, u, s_ref, kappa_ref, uncrossed=True)


def jac_complex_step(f, x, u, s_ref, kappa_ref, h=1e-30):
    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    J = np.zeros((8, 10))
    for j in range(10):
        xc = x.astype(np.complex128)
        uc = u.astype(np.complex128)
        if j < 8:
            xc[j] += 1j * h
        else:
            uc[j - 8] += 1j * h
        J[:, j] = np.imag(f(xc, uc, s_ref, kappa_ref)) / h
    return J


# ---- fdyn10: 15-state Frenet model with wheel speeds (python/models.py:609-801), IMPLICIT residual as the reference writes it ----
b1s, b2s, b3s, c1s, d1s, d2s, e1s, e2s, e3s = -6.75e-6, 1.35e-1, 1.2e-3, 1.86, 1.12e-4, 1.57, -5.38e-6, 1.11e-2, -4.26
BCDs = (b1s * static_weight**2 + b2s * static_weight) * np.exp(-b3s * static_weight)
Cs = c1s
Ds = d1s * static_weight + d2s
Es = e1s * static_weight**2 + e2s * static_weight + e3s
Bs = BCDs / (Cs * Ds)
R_w, I_w, k_d, k_s = 0.20809, 0.3, 0.17, 15.0


def lon_pacejka(s):
    return Ds * np.sin(Cs * np.arctan(Bs * s - Es * (Bs * s - np.arctan(Bs * s))))


def fdyn10_residual(xdot, x, u, s_ref, kappa_ref):
    """Residual vector in the order of python/models.py:747-801 (zero when xdot is the model's derivative)."""
    s, n, psi, v_x, v_y, r, o_FL, o_FR, o_RL, o_RR, tau_FL, tau_FR, tau_RL, tau_RR, delta = x
    u_FL, u_FR, u_RL, u_RR, u_delta = u
    (s_dot, n_dot, psi_dot, v_x_dot, v_y_dot, r_dot, o_FL_dot, o_FR_dot, o_RL_dot, o_RR_dot,
     tau_FL_dot, tau_FR_dot, tau_RL_dot, tau_RR_dot, delta_dot) = xdot
    a_x = v_x_dot - v_y * r
    a_y = v_y_dot + v_x * r
    F_drag = -(C_r0 + C_r1 * v_x + C_r2 * v_x * v_x) * np.tanh(1000 * v_x)
    F_downforce = 0.5 * C_downforce * v_x * v_x
    lon_wt = 0.5 * m * a_x * z_CG / wheelbase
    lat_wt = 0.5 * m * a_y * z_CG / front_axle_track
    F_z_FL = -(static_weight - lon_wt + lat_wt + 0.25 * F_downforce)
    F_z_FR = -(static_weight - lon_wt - lat_wt + 0.25 * F_downforce)
    F_z_RL = -(static_weight + lon_wt + lat_wt + 0.25 * F_downforce)
    F_z_RR = -(static_weight + lon_wt - lat_wt + 0.25 * F_downforce)
    v_x_FL = v_x - 0.5 * front_axle_track * r
    v_x_FR = v_x + 0.5 * front_axle_track * r
    v_y_FL = v_y_FR = v_y + l_F * r
    cd, sd = np.cos(delta), np.sin(delta)
    v_lon_FL, v_lon_FR = cd * v_x_FL + sd * v_y_FL, cd * v_x_FR + sd * v_y_FR
    v_lat_FL, v_lat_FR = -sd * v_x_FL + cd * v_y_FL, -sd * v_x_FR + cd * v_y_FR
    v_lon_RL, v_lon_RR = v_x - 0.5 * rear_axle_track * r, v_x + 0.5 * rear_axle_track * r
    v_lat_RL = v_lat_RR = v_y - l_R * r
    sa = [smooth_abs_nonzero(v) for v in (v_lon_FL, v_lon_FR, v_lon_RL, v_lon_RR)]
    alpha = [np.arctan2(vl, a) for vl, a in zip((v_lat_FL, v_lat_FR, v_lat_RL, v_lat_RR), sa)]
    Fz = (F_z_FL, F_z_FR, F_z_RL, F_z_RR)
    F_lat = [fz * lat_pacejka(al) for fz, al in zip(Fz, alpha)]
    slip = [o * R_w / a - 1.0 for o, a in zip((o_FL, o_FR, o_RL, o_RR), sa)]
    F_lon = [-fz * lon_pacejka(sl) for fz, sl in zip(Fz, slip)]
    kap = kappa_interp(s, s_ref, kappa_ref)
    s_dot_expr = (v_x * np.cos(psi) - v_y * np.sin(psi)) / (1 + kap * n)
    return np.array([
        s_dot - s_dot_expr,
        n_dot - (v_x * np.sin(psi) + v_y * np.cos(psi)),
        psi_dot - (r - kap * s_dot_expr),
        m * a_x - (F_drag + cd * (F_lon[0] + F_lon[1]) - sd * (F_lat[0] + F_lat[1]) + F_lon[2] + F_lon[3]),
        m * a_y - (sd * (F_lon[0] + F_lon[1]) + cd * (F_lat[0] + F_lat[1]) + F_lat[2] + F_lat[3]),
        I_z * r_dot - (
            (F_lon[1] * cd - F_lat[1] * sd) * front_axle_track / 2 + (F_lon[1] * sd + F_lat[1] * cd) * l_F
            - (F_lon[0] * cd - F_lat[0] * sd) * front_axle_track / 2 + (F_lon[0] * sd + F_lat[0] * cd) * l_F
            + F_lon[3] * rear_axle_track / 2 - F_lat[3] * l_R - F_lon[2] * rear_axle_track / 2 - F_lat[2] * l_R
        ),
        I_w * o_FL_dot - (tau_FL - (k_d * o_FL + k_s + R_w * F_lon[0])),
        I_w * o_FR_dot - (tau_FR - (k_d * o_FR + k_s + R_w * F_lon[1])),
        I_w * o_RL_dot - (tau_RL - (k_d * o_RL + k_s + R_w * F_lon[2])),
        I_w * o_RR_dot - (tau_RR - (k_d * o_RR + k_s + R_w * F_lon[3])),
        tau_FL_dot - (u_FL - tau_FL) / t_T,
        tau_FR_dot - (u_FR - tau_FR) / t_T,
        tau_RL_dot - (u_RL - tau_RL) / t_T,
        tau_RR_dot - (u_RR - tau_RR) / t_T,
        delta_dot - (u_delta - delta) / t_delta,
    ])
