/*
 * ihm2_oracle_model.c -- CPU ORACLE (test infrastructure, NOT the product).
 * Vehicle models and the sub-stepped RK4 integrator with forward sensitivities.
 *
 * Follows (reference file:line, relative to /root/reference):
 *   constants        python/constants.py:43-111
 *   smooth_*         python/utils.py:19-32
 *   Pacejka          python/models.py:69-80
 *   fkin6_model      python/models.py:232-307
 *   fdyn6_model      python/models.py:455-606 (implicit residual; solved here for xdot)
 *   kappa interpolant python/models.py:290-292 (casadi "linear" interpolant on the grid p[:n], values p[n:])
 *   RK4 tableau      dpc/main.py:87-97 (the correct one; python/mpc.py:135-139 has quirk Q2)
 *
 * The model is written ONCE over C99 `double complex` with first-order-exact elementary
 * functions (cs_*): the real part is the model, the imaginary part carries one directional
 * derivative (complex-step / dual-number evaluation, step 1e-30), so the Jacobian of either
 * model is exact to rounding without hand differentiation. fkin6 additionally has a
 * hand-derived analytic Jacobian (fast path used by the CPU baseline); tests compare the two.
 */
#include "ihm2_oracle.h"

#include <complex.h>
#include <math.h>
#include <string.h>

/* ---- constants (python/constants.py:43-111) ---- */
static const double g_ = 9.81;
static const double m_ = 230.0;
static const double I_z = 137.583;
static const double z_CG = 0.295;
static const double axle_track = 1.24;
static const double l_R = 0.7853;
static const double l_F = 0.7853;
static const double wheelbase = 1.5706;
static const double C_m0 = 4.950;
static const double C_r0 = 297.030;
static const double C_r1 = 16.665;
static const double C_r2 = 0.6784;
static const double b1a = 3.79e1, b2a = 5.28e2, c1a = 1.57, d1a = -2.03e-4, d2a = 1.77,
                    e1a = -2.24e-3, e2a = 1.81;
static const double t_T = 1e-3;
static const double t_delta = 0.02;
static const double C_downforce = 3.96864;
static const double K_tv = 300.0;

typedef double complex cplx;

/* first-order-exact complex elementary functions: f(a+ib) = f(a) + i b f'(a) */
static inline cplx cs_make(double a, double b) { return a + b * I; }
static inline cplx cs_sin(cplx z) { double a = creal(z); return cs_make(sin(a), cimag(z) * cos(a)); }
static inline cplx cs_cos(cplx z) { double a = creal(z); return cs_make(cos(a), -cimag(z) * sin(a)); }
static inline cplx cs_tan(cplx z) { double a = creal(z), t = tan(a); return cs_make(t, cimag(z) * (1.0 + t * t)); }
static inline cplx cs_atan(cplx z) { double a = creal(z); return cs_make(atan(a), cimag(z) / (1.0 + a * a)); }
static inline cplx cs_tanh(cplx z) { double a = creal(z), t = tanh(a); return cs_make(t, cimag(z) * (1.0 - t * t)); }
static inline cplx cs_exp(cplx z) { double e = exp(creal(z)); return cs_make(e, cimag(z) * e); }
static inline cplx cs_sqrt(cplx z) { double r = sqrt(creal(z)); return cs_make(r, cimag(z) / (2.0 * r)); }
/* atan2(y, x) with x > 0 (x is always smooth_abs_nonzero(.) > 0 at the call sites) */
static inline cplx cs_atan2_pos(cplx y, cplx x) { return cs_atan(y / x); }
/* general atan2 (any quadrant): d atan2(y, x) = (x dy - y dx) / (x^2 + y^2) */
static inline cplx cs_atan2(cplx y, cplx x)
{
    double yr = creal(y), xr = creal(x);
    return cs_make(atan2(yr, xr), (xr * cimag(y) - yr * cimag(x)) / (xr * xr + yr * yr));
}

/* python/utils.py:23-32 */
static inline cplx cs_smooth_sgn(cplx x) { return cs_tanh(10.0 * x); }
static inline cplx cs_smooth_abs_nonzero(cplx x) { return cs_smooth_sgn(x) * x + 1e-6 * cs_exp(-x * x); }

/* python/constants.py:85-95 and python/models.py:69-74 */
static void pacejka_lat_params(double *Ba, double *Ca, double *Da, double *Ea)
{
    double static_weight = 0.5 * m_ * g_ * l_F / wheelbase;
    double BCDa = b1a * sin(2.0 * atan(static_weight / b2a));
    *Ca = c1a;
    *Da = d1a * static_weight + d2a;
    *Ea = e1a * static_weight + e2a;
    *Ba = BCDa / (*Ca * *Da);
}
static inline cplx cs_lat_pacejka(cplx alpha)
{
    double Ba, Ca, Da, Ea;
    pacejka_lat_params(&Ba, &Ca, &Da, &Ea);
    cplx Bx = Ba * alpha;
    return Da * cs_sin(Ca * cs_atan(Bx - Ea * (Bx - cs_atan(Bx))));
}

/* ---- kappa(s): piecewise-linear on the grid, linear extrapolation outside ---- */
static int find_segment(const double *s_ref, int n, double s)
{
    /* largest i in [0, n-2] with s_ref[i] <= s (binary search, "exact" lookup mode) */
    int lo = 0, hi = n - 1;
    if (!(s >= s_ref[0])) return 0;
    if (s >= s_ref[n - 1]) return n - 2;
    while (hi - lo > 1) {
        int mid = (lo + hi) / 2;
        if (s_ref[mid] <= s) lo = mid; else hi = mid;
    }
    return lo;
}

double orc_kappa(const double *s_ref, const double *kappa_ref, int nknots, double s, double *dkappa_ds)
{
    int i = find_segment(s_ref, nknots, s);
    double slope = (kappa_ref[i + 1] - kappa_ref[i]) / (s_ref[i + 1] - s_ref[i]);
    if (dkappa_ds) *dkappa_ds = slope;
    return kappa_ref[i] + slope * (s - s_ref[i]);
}

static inline cplx cs_kappa(const double *s_ref, const double *kappa_ref, int nknots, cplx s)
{
    double dk, k = orc_kappa(s_ref, kappa_ref, nknots, creal(s), &dk);
    return cs_make(k, cimag(s) * dk);
}

/* ---- fkin6 (python/models.py:232-307) ---- */
static void fkin6_cs(const cplx *x, const cplx *u, const double *s_ref, const double *kappa_ref, int nk, cplx *f)
{
    const double rwd = l_R / wheelbase; /* rear_weight_distribution, constants.py:54 */
    cplx s = x[0], n = x[1], psi = x[2], v_x = x[3], v_y = x[4], r = x[5], T = x[6], delta = x[7];
    cplx u_T = u[0], u_delta = u[1];
    /* actuator dynamics :251-252 */
    cplx delta_dot = (u_delta - delta) / t_delta;
    cplx T_dot = (u_T - T) / t_T;
    /* longitudinal dynamics :255-258 */
    cplx F_motor = C_m0 * T;
    cplx F_drag = -(C_r0 + C_r1 * v_x + C_r2 * v_x * v_x) * cs_smooth_sgn(v_x);
    cplx F_Rx = 0.5 * F_motor + F_drag;
    cplx F_Fx = 0.5 * F_motor;
    /* lateral dynamics :261-284 */
    cplx tandelta = cs_tan(delta);
    cplx beta = cs_atan(rwd * tandelta);
    cplx cos_beta = cs_cos(beta), sin_beta = cs_sin(beta);
    cplx beta_dot = rwd * (1.0 + tandelta * tandelta) / (1.0 + rwd * rwd * tandelta * tandelta) * delta_dot;
    /* accelerations :287 */
    cplx v_dot = (F_Rx * cos_beta + F_Fx * cs_cos(delta - beta)) / m_;
    /* Frenet kinematics :290-296 */
    cplx kap = cs_kappa(s_ref, kappa_ref, nk, s);
    cplx s_dot = (v_x * cs_cos(psi) - v_y * cs_sin(psi)) / (1.0 + kap * n);
    cplx v_y_dot = v_dot * sin_beta + beta_dot * v_x;
    f[0] = s_dot;
    f[1] = v_x * cs_sin(psi) + v_y * cs_cos(psi);
    f[2] = r - kap * s_dot;
    f[3] = v_dot * cos_beta - beta_dot * v_y;
    f[4] = v_y_dot;
    f[5] = l_R * v_y_dot - beta_dot; /* quirk Q4: as written in models.py:304 */
    f[6] = T_dot;
    f[7] = delta_dot;
}

/* ---- lateral acceleration of the kinematic model: the nonlinear row `a_lat` of old/generate_acaods_interface.py:198-209 ----
 * definition old/generate_acaods_interface.py:266-271 = old/scripts/gen_mpc.py:182-184, on the forces and the slip angle of the live
 * kinematic model (python/models.py:255-263) */
static cplx alat_cs(const cplx *x)
{
    const double rwd = l_R / wheelbase;
    cplx v_x = x[3], v_y = x[4], T = x[6], delta = x[7];
    cplx F_motor = C_m0 * T;
    cplx F_drag = -(C_r0 + C_r1 * v_x + C_r2 * v_x * v_x) * cs_smooth_sgn(v_x);
    cplx F_Rx = 0.5 * F_motor + F_drag;
    cplx F_Fx = 0.5 * F_motor;
    cplx beta = cs_atan(rwd * cs_tan(delta));
    cplx sinbeta = cs_sin(beta);
    return (-F_Rx * sinbeta + F_Fx * cs_sin(delta - beta)) / m_ + (v_x * v_x + v_y * v_y) * sinbeta / l_R;
}

double orc_alat(const double *x, double *grad)
{
    const double hstep = 1e-30;
    cplx xc[ORC_NX];
    for (int i = 0; i < ORC_NX; i++) xc[i] = x[i];
    const double val = creal(alat_cs(xc));
    if (grad)
        for (int j = 0; j < ORC_NX; j++) {
            xc[j] = cs_make(x[j], hstep);
            grad[j] = cimag(alat_cs(xc)) / hstep;
            xc[j] = x[j];
        }
    return val;
}

/* ---- fdyn6 (python/models.py:455-606), explicit form via the 2x2 solve of SURVEY.md C.2 ---- */
static void fdyn6_cs(const cplx *x, const cplx *u, const double *s_ref, const double *kappa_ref, int nk, cplx *f, int uncrossed)
{
    const double rwd = l_R / wheelbase;
    cplx s = x[0], n = x[1], psi = x[2], v_x = x[3], v_y = x[4], r = x[5], T = x[6], delta = x[7];
    cplx u_T = u[0], u_delta = u[1];
    cplx sd = cs_sin(delta), cd = cs_cos(delta);

    cplx F_downforce = 0.5 * C_downforce * v_x * v_x;            /* :487 */
    double static_weight = 0.5 * m_ * g_ * l_F / wheelbase;      /* :488 */
    const double cx = 0.5 * m_ * z_CG / wheelbase;               /* longitudinal transfer per a_x :489 */
    const double cy = 0.5 * m_ * z_CG / axle_track;              /* lateral transfer per a_y :490 */
    cplx base = static_weight + 0.25 * F_downforce;
    /* F_z,ij = -(base + sx*cx*a_x + sy*cy*a_y), order FL, FR, RL, RR (:492-515) */
    static const double sx[4] = {-1.0, -1.0, 1.0, 1.0};
    static const double sy[4] = {1.0, -1.0, 1.0, -1.0};

    /* wheel velocities :518-529 */
    cplx v_x_FL = v_x - 0.5 * axle_track * r, v_x_FR = v_x + 0.5 * axle_track * r;
    cplx v_y_FL = v_y + l_F * r, v_y_FR = v_y + l_F * r;
    cplx v_lon_FL = cd * v_x_FL + sd * v_y_FL, v_lon_FR = cd * v_x_FR + sd * v_y_FR;
    cplx v_lat_FL = -sd * v_x_FL + cd * v_y_FL, v_lat_FR = -sd * v_x_FR + cd * v_y_FR;
    cplx v_lon_RL = v_x - 0.5 * axle_track * r, v_lon_RR = v_x + 0.5 * axle_track * r;
    cplx v_lat_RL = v_y - l_R * r, v_lat_RR = v_y - l_R * r;
    /* slip angles :531-540 */
    cplx alpha_FL = cs_atan2_pos(v_lat_FL, cs_smooth_abs_nonzero(v_lon_FL));
    cplx alpha_FR = cs_atan2_pos(v_lat_FR, cs_smooth_abs_nonzero(v_lon_FR));
    cplx alpha_RL = cs_atan2_pos(v_lat_RL, cs_smooth_abs_nonzero(v_lon_RL));
    cplx alpha_RR = cs_atan2_pos(v_lat_RR, cs_smooth_abs_nonzero(v_lon_RR));
    /* lateral force per unit F_z, with the crossed indices of :543-546 (quirk Q3) */
    cplx glat[4] = {cs_lat_pacejka(alpha_RR), cs_lat_pacejka(alpha_RL), cs_lat_pacejka(alpha_FR),
                    cs_lat_pacejka(alpha_FL)};
    if (uncrossed) { /* ORC_MODEL_FDYN6U: every wheel with its own slip angle (named deviation from the reference) */
        glat[0] = cs_lat_pacejka(alpha_FL); glat[1] = cs_lat_pacejka(alpha_FR);
        glat[2] = cs_lat_pacejka(alpha_RL); glat[3] = cs_lat_pacejka(alpha_RR);
    }
    /* longitudinal :549-560 */
    cplx F_drag = -(C_r0 + C_r1 * v_x + C_r2 * v_x * v_x) * cs_smooth_sgn(v_x);
    cplx beta = cs_atan(rwd * cs_tan(delta));
    cplx r_kin = cs_sqrt(v_x * v_x + v_y * v_y) * cs_sin(beta) / l_R;
    cplx delta_tau = K_tv * (r_kin - r);
    cplx denom = -m_ * g_ - 0.25 * F_downforce;
    cplx glon[4] = {C_m0 * (T - delta_tau) / denom, C_m0 * (T + delta_tau) / denom,
                    C_m0 * (T - delta_tau) / denom, C_m0 * (T + delta_tau) / denom};

    /* force-direction coefficients: F_x = sum cxk[k]*F_z[k] + F_drag, F_y = sum cyk[k]*F_z[k],
       M_z = sum czk[k]*F_z[k]  (:578-603) */
    cplx cxk[4], cyk[4], czk[4];
    cxk[0] = glon[0] * cd - glat[0] * sd;  cyk[0] = glon[0] * sd + glat[0] * cd; /* FL */
    cxk[1] = glon[1] * cd - glat[1] * sd;  cyk[1] = glon[1] * sd + glat[1] * cd; /* FR */
    cxk[2] = glon[2];                      cyk[2] = glat[2];                     /* RL */
    cxk[3] = glon[3];                      cyk[3] = glat[3];                     /* RR */
    czk[0] = -cxk[0] * (0.5 * axle_track) + cyk[0] * l_F;
    czk[1] = cxk[1] * (0.5 * axle_track) + cyk[1] * l_F;
    czk[2] = -glon[2] * (0.5 * axle_track) - glat[2] * l_R;
    czk[3] = glon[3] * (0.5 * axle_track) - glat[3] * l_R;

    /* m a_x = X0 + Xx a_x + Xy a_y ;  m a_y = Y0 + Yx a_x + Yy a_y */
    cplx X0 = F_drag, Xx = 0, Xy = 0, Y0 = 0, Yx = 0, Yy = 0;
    for (int k = 0; k < 4; k++) {
        X0 -= cxk[k] * base;  Xx -= cxk[k] * (sx[k] * cx);  Xy -= cxk[k] * (sy[k] * cy);
        Y0 -= cyk[k] * base;  Yx -= cyk[k] * (sx[k] * cx);  Yy -= cyk[k] * (sy[k] * cy);
    }
    cplx a11 = m_ - Xx, a12 = -Xy, a21 = -Yx, a22 = m_ - Yy;
    cplx det = a11 * a22 - a12 * a21;
    cplx a_x = (X0 * a22 - a12 * Y0) / det;
    cplx a_y = (a11 * Y0 - a21 * X0) / det;
    cplx Mz = 0;
    for (int k = 0; k < 4; k++) {
        cplx Fz = -(base + sx[k] * cx * a_x + sy[k] * cy * a_y);
        Mz += czk[k] * Fz;
    }
    /* Frenet kinematics :563-571 */
    cplx kap = cs_kappa(s_ref, kappa_ref, nk, s);
    cplx s_dot = (v_x * cs_cos(psi) - v_y * cs_sin(psi)) / (1.0 + kap * n);
    f[0] = s_dot;
    f[1] = v_x * cs_sin(psi) + v_y * cs_cos(psi);
    f[2] = r - kap * s_dot;
    f[3] = a_x + v_y * r; /* a_x = v_x_dot - v_y r  (:483) */
    f[4] = a_y - v_x * r; /* a_y = v_y_dot + v_x r  (:484) */
    f[5] = Mz / I_z;
    f[6] = (u_T - T) / t_T;
    f[7] = (u_delta - delta) / t_delta;
}

/* ---- Cartesian plants of the ROS simulation node (src/ihm2/src/sim_node.cpp:197-257): x = (X, Y, phi, v_x, v_y, r, T, delta) ---- */
/* kin6 (python/models.py:168-229): the fkin6 force model with Cartesian kinematics and r_dot = v_y_dot / l_R (:226) */
static void kin6_cs(const cplx *x, const cplx *u, cplx *f)
{
    const double rwd = l_R / wheelbase;
    cplx phi = x[2], v_x = x[3], v_y = x[4], r = x[5], T = x[6], delta = x[7];
    cplx delta_dot = (u[1] - delta) / t_delta, T_dot = (u[0] - T) / t_T;
    cplx F_motor = C_m0 * T;
    cplx F_drag = -(C_r0 + C_r1 * v_x + C_r2 * v_x * v_x) * cs_smooth_sgn(v_x);
    cplx F_Rx = 0.5 * F_motor + F_drag, F_Fx = 0.5 * F_motor;
    cplx tandelta = cs_tan(delta);
    cplx den = cs_sqrt(1.0 + rwd * rwd * tandelta * tandelta);
    cplx beta = cs_atan(rwd * tandelta);
    cplx sinbeta = rwd * tandelta / den, cosbeta = 1.0 / den;         /* :196-207, algebraic */
    cplx beta_dot = rwd * (1.0 + tandelta * tandelta) / (1.0 + rwd * rwd * tandelta * tandelta) * delta_dot;
    cplx v_dot = (F_Rx * cosbeta + F_Fx * cs_cos(delta - beta)) / m_;
    cplx v_y_dot = v_dot * sinbeta + beta_dot * v_x;
    f[0] = v_x * cs_cos(phi) - v_y * cs_sin(phi);
    f[1] = v_x * cs_sin(phi) + v_y * cs_cos(phi);
    f[2] = r;
    f[3] = v_dot * cosbeta - beta_dot * v_y;
    f[4] = v_y_dot;
    f[5] = v_y_dot / l_R;
    f[6] = T_dot;
    f[7] = delta_dot;
}

/* dyn6 (python/models.py:310-452), explicit form: the same 2x2 solve for (a_x, a_y) as fdyn6.  Differences to fdyn6 as the
 * reference writes them: slip angles atan2(v_y_w, v_x_w) - delta in the body frame (:376-379), every wheel on its own slip
 * angle (:380-395), drag with tanh(1000 v_x) (:398), r_kin = v_x sin(beta) / l_R (:409). */
static void dyn6_cs(const cplx *x, const cplx *u, cplx *f)
{
    const double rwd = l_R / wheelbase;
    cplx phi = x[2], v_x = x[3], v_y = x[4], r = x[5], T = x[6], delta = x[7];
    cplx sd = cs_sin(delta), cd = cs_cos(delta);
    cplx F_downforce = 0.5 * C_downforce * v_x * v_x;
    const double static_weight = 0.5 * m_ * g_ * l_F / wheelbase;
    const double cx = 0.5 * m_ * z_CG / wheelbase, cy = 0.5 * m_ * z_CG / axle_track;
    cplx base = static_weight + 0.25 * F_downforce;
    static const double sx[4] = {-1.0, -1.0, 1.0, 1.0};
    static const double sy[4] = {1.0, -1.0, 1.0, -1.0};
    cplx v_x_L = v_x - 0.5 * axle_track * r, v_x_R = v_x + 0.5 * axle_track * r;
    cplx v_y_F = v_y + l_F * r, v_y_R = v_y - l_R * r;
    cplx alpha[4] = {cs_atan2(v_y_F, v_x_L) - delta, cs_atan2(v_y_F, v_x_R) - delta, cs_atan2(v_y_R, v_x_L), cs_atan2(v_y_R, v_x_R)};
    cplx glat[4], glon[4];
    for (int k = 0; k < 4; k++) glat[k] = cs_lat_pacejka(alpha[k]);
    cplx F_drag = -(C_r0 + C_r1 * v_x + C_r2 * v_x * v_x) * cs_tanh(1000.0 * v_x);
    cplx tandelta = cs_tan(delta);
    cplx sinbeta = rwd * tandelta / cs_sqrt(1.0 + rwd * rwd * tandelta * tandelta);
    cplx delta_tau = K_tv * (v_x * sinbeta / l_R - r);
    cplx denom = -m_ * g_ - 0.25 * F_downforce;
    glon[0] = C_m0 * (T - delta_tau) / denom; glon[1] = C_m0 * (T + delta_tau) / denom;
    glon[2] = glon[0]; glon[3] = glon[1];
    cplx cxk[4], cyk[4], czk[4];
    cxk[0] = glon[0] * cd - glat[0] * sd;  cyk[0] = glon[0] * sd + glat[0] * cd;
    cxk[1] = glon[1] * cd - glat[1] * sd;  cyk[1] = glon[1] * sd + glat[1] * cd;
    cxk[2] = glon[2];                      cyk[2] = glat[2];
    cxk[3] = glon[3];                      cyk[3] = glat[3];
    czk[0] = -cxk[0] * (0.5 * axle_track) + cyk[0] * l_F;
    czk[1] = cxk[1] * (0.5 * axle_track) + cyk[1] * l_F;
    czk[2] = -glon[2] * (0.5 * axle_track) - glat[2] * l_R;
    czk[3] = glon[3] * (0.5 * axle_track) - glat[3] * l_R;
    cplx X0 = F_drag, Xx = 0, Xy = 0, Y0 = 0, Yx = 0, Yy = 0;
    for (int k = 0; k < 4; k++) {
        X0 -= cxk[k] * base;  Xx -= cxk[k] * (sx[k] * cx);  Xy -= cxk[k] * (sy[k] * cy);
        Y0 -= cyk[k] * base;  Yx -= cyk[k] * (sx[k] * cx);  Yy -= cyk[k] * (sy[k] * cy);
    }
    cplx a11 = m_ - Xx, a12 = -Xy, a21 = -Yx, a22 = m_ - Yy;
    cplx det = a11 * a22 - a12 * a21;
    cplx a_x = (X0 * a22 - a12 * Y0) / det, a_y = (a11 * Y0 - a21 * X0) / det;
    cplx Mz = 0;
    for (int k = 0; k < 4; k++) Mz += czk[k] * -(base + sx[k] * cx * a_x + sy[k] * cy * a_y);
    f[0] = v_x * cs_cos(phi) - v_y * cs_sin(phi);
    f[1] = v_x * cs_sin(phi) + v_y * cs_cos(phi);
    f[2] = r;
    f[3] = a_x + v_y * r;
    f[4] = a_y - v_x * r;
    f[5] = Mz / I_z;
    f[6] = (u[0] - T) / t_T;
    f[7] = (u[1] - delta) / t_delta;
}

static void f_cs(int model, const cplx *x, const cplx *u, const double *s_ref, const double *kappa_ref, int nk, cplx *f)
{
    if (model == ORC_MODEL_KIN6) kin6_cs(x, u, f);
    else if (model == ORC_MODEL_DYN6) dyn6_cs(x, u, f);
    else if (model == ORC_MODEL_FDYN6 || model == ORC_MODEL_FDYN6U) fdyn6_cs(x, u, s_ref, kappa_ref, nk, f, model == ORC_MODEL_FDYN6U);
    else fkin6_cs(x, u, s_ref, kappa_ref, nk, f);
}

void orc_f(int model, const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots, double *xdot)
{
    cplx xc[ORC_NX], uc[ORC_NU], fc[ORC_NX];
    for (int i = 0; i < ORC_NX; i++) xc[i] = x[i];
    for (int i = 0; i < ORC_NU; i++) uc[i] = u[i];
    f_cs(model, xc, uc, s_ref, kappa_ref, nknots, fc);
    for (int i = 0; i < ORC_NX; i++) xdot[i] = creal(fc[i]);
}

/* Jacobian by complex-step evaluation of the same formulas (exact to rounding). */
void orc_jac_cs(int model, const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots, double *xdot, double *J)
{
    const double h = 1e-30;
    cplx xc[ORC_NX], uc[ORC_NU], fc[ORC_NX];
    for (int j = 0; j < ORC_NZ; j++) {
        for (int i = 0; i < ORC_NX; i++) xc[i] = x[i];
        for (int i = 0; i < ORC_NU; i++) uc[i] = u[i];
        if (j < ORC_NX) xc[j] += h * I; else uc[j - ORC_NX] += h * I;
        f_cs(model, xc, uc, s_ref, kappa_ref, nknots, fc);
        for (int i = 0; i < ORC_NX; i++) J[i * ORC_NZ + j] = cimag(fc[i]) / h;
    }
    if (xdot) for (int i = 0; i < ORC_NX; i++) xdot[i] = creal(fc[i]);
}

/* Hand-derived analytic Jacobian of fkin6 (structure: SURVEY.md Appendix C.1). */
static void fkin6_jac_analytic(const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nk, double *f, double *J)
{
    const double c = l_R / wheelbase;
    double s = x[0], n = x[1], psi = x[2], v_x = x[3], v_y = x[4], r = x[5], T = x[6], delta = x[7];
    double u_T = u[0], u_delta = u[1];
    memset(J, 0, sizeof(double) * ORC_NX * ORC_NZ);
#define JJ(i, j) J[(i) * ORC_NZ + (j)]
    double delta_dot = (u_delta - delta) / t_delta;
    double T_dot = (u_T - T) / t_T;
    double F_motor = C_m0 * T;
    double sg = tanh(10.0 * v_x);
    double poly = C_r0 + C_r1 * v_x + C_r2 * v_x * v_x;
    double F_drag = -poly * sg;
    double dFdrag = -(C_r1 + 2.0 * C_r2 * v_x) * sg - poly * 10.0 * (1.0 - sg * sg);
    double F_Rx = 0.5 * F_motor + F_drag, F_Fx = 0.5 * F_motor;
    double td = tan(delta);
    double beta = atan(c * td);
    double cb = cos(beta), sb = sin(beta);
    double den = 1.0 + c * c * td * td;
    double bp = c * (1.0 + td * td) / den;                                  /* d beta / d delta */
    double bpp = 2.0 * c * td * (1.0 + td * td) * (1.0 - c * c) / (den * den); /* d2 beta / d delta2 */
    double beta_dot = bp * delta_dot;
    double dbd_ddelta = bpp * delta_dot - bp / t_delta;
    double dbd_du = bp / t_delta;
    double cdb = cos(delta - beta), sdb = sin(delta - beta);
    double v_dot = (F_Rx * cb + F_Fx * cdb) / m_;
    double dv_dvx = dFdrag * cb / m_;
    double dv_dT = 0.5 * C_m0 * (cb + cdb) / m_;
    double dv_dd = (-F_Rx * sb * bp - F_Fx * sdb * (1.0 - bp)) / m_;
    double dk, kap = orc_kappa(s_ref, kappa_ref, nk, s, &dk);
    double cp = cos(psi), sp = sin(psi);
    double num = v_x * cp - v_y * sp;
    double dn = 1.0 + kap * n;
    double s_dot = num / dn;
    double v_y_dot = v_dot * sb + beta_dot * v_x;
    f[0] = s_dot;
    f[1] = v_x * sp + v_y * cp;
    f[2] = r - kap * s_dot;
    f[3] = v_dot * cb - beta_dot * v_y;
    f[4] = v_y_dot;
    f[5] = l_R * v_y_dot - beta_dot;
    f[6] = T_dot;
    f[7] = delta_dot;
    /* row 0 */
    JJ(0, 0) = -num / (dn * dn) * dk * n;
    JJ(0, 1) = -num / (dn * dn) * kap;
    JJ(0, 2) = (-v_x * sp - v_y * cp) / dn;
    JJ(0, 3) = cp / dn;
    JJ(0, 4) = -sp / dn;
    /* row 1 */
    JJ(1, 2) = v_x * cp - v_y * sp;
    JJ(1, 3) = sp;
    JJ(1, 4) = cp;
    /* row 2 */
    JJ(2, 0) = -dk * s_dot - kap * JJ(0, 0);
    JJ(2, 1) = -kap * JJ(0, 1);
    JJ(2, 2) = -kap * JJ(0, 2);
    JJ(2, 3) = -kap * JJ(0, 3);
    JJ(2, 4) = -kap * JJ(0, 4);
    JJ(2, 5) = 1.0;
    /* row 3: v_dot cb - beta_dot v_y */
    JJ(3, 3) = dv_dvx * cb;
    JJ(3, 4) = -beta_dot;
    JJ(3, 6) = dv_dT * cb;
    JJ(3, 7) = dv_dd * cb - v_dot * sb * bp - dbd_ddelta * v_y;
    JJ(3, 9) = -dbd_du * v_y;
    /* row 4: v_dot sb + beta_dot v_x */
    JJ(4, 3) = dv_dvx * sb + beta_dot;
    JJ(4, 6) = dv_dT * sb;
    JJ(4, 7) = dv_dd * sb + v_dot * cb * bp + dbd_ddelta * v_x;
    JJ(4, 9) = dbd_du * v_x;
    /* row 5: l_R v_y_dot - beta_dot */
    JJ(5, 3) = l_R * JJ(4, 3);
    JJ(5, 6) = l_R * JJ(4, 6);
    JJ(5, 7) = l_R * JJ(4, 7) - dbd_ddelta;
    JJ(5, 9) = l_R * JJ(4, 9) - dbd_du;
    /* rows 6, 7 */
    JJ(6, 6) = -1.0 / t_T;
    JJ(6, 8) = 1.0 / t_T;
    JJ(7, 7) = -1.0 / t_delta;
    JJ(7, 9) = 1.0 / t_delta;
#undef JJ
}

void orc_jac(int model, const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots, double *xdot, double *J)
{
    double f[ORC_NX];
    if (model == ORC_MODEL_FKIN6) {
        fkin6_jac_analytic(x, u, s_ref, kappa_ref, nknots, f, J);
        if (xdot) memcpy(xdot, f, sizeof f);
    } else {
        orc_jac_cs(model, x, u, s_ref, kappa_ref, nknots, xdot, J);
    }
}

/* ---- RK4 x M with forward sensitivities (internal numerical differentiation) ----
 * S = d x_m / d (x_0, u) (8x10), S_0 = [I 0]. Stage i: X_i = x + a_i h K_{i-1},
 * dX_i = S + a_i h dK_{i-1}, dK_i = Jx(X_i) dX_i + [0 | Ju(X_i)].
 * integrator = ORC_INTEG_RK4; the collocation integrators are irk_core below. */
static void stage_eval(int model, int with_sens, const double *X, const double *u, const double *s_ref, const double *kappa_ref, int nk,
                       const double *dX /*8x10*/, double *K, double *dK /*8x10*/)
{
    if (!with_sens) { orc_f(model, X, u, s_ref, kappa_ref, nk, K); return; }
    double J[ORC_NX * ORC_NZ];
    orc_jac(model, X, u, s_ref, kappa_ref, nk, K, J);
    for (int i = 0; i < ORC_NX; i++)
        for (int j = 0; j < ORC_NZ; j++) {
            double acc = (j >= ORC_NX) ? J[i * ORC_NZ + j] : 0.0;
            for (int l = 0; l < ORC_NX; l++) acc += J[i * ORC_NZ + l] * dX[l * ORC_NZ + j];
            dK[i * ORC_NZ + j] = acc;
        }
}

#include "irk_tableaux.h"

/* dense LU with partial pivoting, in place; returns 0 if singular */
static int lu_factor(int n, double *Mx, int *piv)
{
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++) if (fabs(Mx[r * n + c]) > fabs(Mx[p * n + c])) p = r;
        piv[c] = p;
        if (Mx[p * n + c] == 0.0) return 0;
        if (p != c) for (int j = 0; j < n; j++) { double t = Mx[c * n + j]; Mx[c * n + j] = Mx[p * n + j]; Mx[p * n + j] = t; }
        for (int r = c + 1; r < n; r++) {
            const double l = Mx[r * n + c] / Mx[c * n + c];
            Mx[r * n + c] = l;
            for (int j = c + 1; j < n; j++) Mx[r * n + j] -= l * Mx[c * n + j];
        }
    }
    return 1;
}
static void lu_solve(int n, const double *Mx, const int *piv, double *rhs, int nrhs)
{
    /* all row interchanges first (the stored multipliers were swapped along with later pivots), then L, then U */
    for (int c = 0; c < n; c++)
        if (piv[c] != c) for (int j = 0; j < nrhs; j++) { double t = rhs[c * nrhs + j]; rhs[c * nrhs + j] = rhs[piv[c] * nrhs + j]; rhs[piv[c] * nrhs + j] = t; }
    for (int c = 0; c < n; c++)
        for (int r = c + 1; r < n; r++) for (int j = 0; j < nrhs; j++) rhs[r * nrhs + j] -= Mx[r * n + c] * rhs[c * nrhs + j];
    for (int c = n - 1; c >= 0; c--) {
        for (int j = 0; j < nrhs; j++) rhs[c * nrhs + j] /= Mx[c * n + c];
        for (int r = 0; r < c; r++) for (int j = 0; j < nrhs; j++) rhs[r * nrhs + j] -= Mx[r * n + c] * rhs[c * nrhs + j];
    }
}

/* ---- 4-stage collocation (IRK) x M steps with forward sensitivities (acados IRK, python/main.py:234-236,395-400) ----
 * Stage values K_i solve K_i = f(x + h sum_j A_ij K_j, u): ORC_IRK_NEWTON_ITER Newton iterations from K = 0, Jacobian
 * I - h (A (x) J) re-evaluated in each (32 x 32, dense LU); x+ = x + h sum_i b_i K_i.  Sensitivities by the implicit-function
 * theorem at the final stage values: (I - h A (x) J) dK = [J_x S + (0 | J_u)]_i, S+ = S + h sum_i b_i dK_i. */
static void irk_core(int model, int integrator, int with_sens, const double *x0, const double *u, const double *s_ref, const double *kappa_ref, int nk,
                     double dt, int M, double *xn, double *S)
{
    const double (*At)[4] = (integrator == ORC_INTEG_IRK_RADAU4) ? IRK_RADAU4_A : IRK_GL4_A;
    const double *bt = (integrator == ORC_INTEG_IRK_RADAU4) ? IRK_RADAU4_b : IRK_GL4_b;
    enum { NS4 = 4, NK = 4 * ORC_NX };
    const double h = dt / M;
    double x[ORC_NX];
    memcpy(x, x0, sizeof x);
    if (with_sens) {
        memset(S, 0, sizeof(double) * ORC_NX * ORC_NZ);
        for (int i = 0; i < ORC_NX; i++) S[i * ORC_NZ + i] = 1.0;
    }
    for (int m = 0; m < M; m++) {
        double K[NS4][ORC_NX], F[NS4][ORC_NX], J[NS4][ORC_NX * ORC_NZ], Mx[NK * NK];
        int piv[NK];
        memset(K, 0, sizeof K);
        for (int it = 0; it <= ORC_IRK_NEWTON_ITER; it++) {
            const int last = it == ORC_IRK_NEWTON_ITER;      /* the last pass only evaluates the Jacobians the sensitivities need */
            if (last && !with_sens) break;
            for (int i = 0; i < NS4; i++) {
                double X[ORC_NX];
                for (int a = 0; a < ORC_NX; a++) {
                    double acc = x[a];
                    for (int j = 0; j < NS4; j++) acc += h * At[i][j] * K[j][a];
                    X[a] = acc;
                }
                orc_jac(model, X, u, s_ref, kappa_ref, nk, F[i], J[i]);
            }
            for (int i = 0; i < NS4; i++)
                for (int a = 0; a < ORC_NX; a++)
                    for (int j = 0; j < NS4; j++)
                        for (int b = 0; b < ORC_NX; b++)
                            Mx[(i * ORC_NX + a) * NK + j * ORC_NX + b] = ((i == j && a == b) ? 1.0 : 0.0) - h * At[i][j] * J[i][a * ORC_NZ + b];
            if (!lu_factor(NK, Mx, piv)) { for (int a = 0; a < ORC_NX; a++) x[a] = NAN; break; }
            if (last) break;
            double r[NK];
            for (int i = 0; i < NS4; i++) for (int a = 0; a < ORC_NX; a++) r[i * ORC_NX + a] = -(K[i][a] - F[i][a]);
            lu_solve(NK, Mx, piv, r, 1);
            for (int i = 0; i < NS4; i++) for (int a = 0; a < ORC_NX; a++) K[i][a] += r[i * ORC_NX + a];
        }
        if (with_sens) {
            double R[NK * ORC_NZ];
            for (int i = 0; i < NS4; i++)
                for (int a = 0; a < ORC_NX; a++)
                    for (int c = 0; c < ORC_NZ; c++) {
                        double acc = (c >= ORC_NX) ? J[i][a * ORC_NZ + c] : 0.0;
                        for (int l = 0; l < ORC_NX; l++) acc += J[i][a * ORC_NZ + l] * S[l * ORC_NZ + c];
                        R[(i * ORC_NX + a) * ORC_NZ + c] = acc;
                    }
            lu_solve(NK, Mx, piv, R, ORC_NZ);
            for (int a = 0; a < ORC_NX; a++)
                for (int c = 0; c < ORC_NZ; c++) {
                    double acc = S[a * ORC_NZ + c];
                    for (int i = 0; i < NS4; i++) acc += h * bt[i] * R[(i * ORC_NX + a) * ORC_NZ + c];
                    S[a * ORC_NZ + c] = acc;
                }
        }
        for (int a = 0; a < ORC_NX; a++) {
            double acc = x[a];
            for (int i = 0; i < NS4; i++) acc += h * bt[i] * K[i][a];
            x[a] = acc;
        }
    }
    memcpy(xn, x, sizeof x);
}

static void rk4_core(int model, int integrator, int with_sens, const double *x0, const double *u, const double *s_ref, const double *kappa_ref, int nk,
                     double dt, int M, double *xn, double *S)
{
    if (integrator != ORC_INTEG_RK4) { irk_core(model, integrator, with_sens, x0, u, s_ref, kappa_ref, nk, dt, M, xn, S); return; }
    const double h = dt / M;
    static const double a[4] = {0.0, 0.5, 0.5, 1.0};
    static const double w[4] = {1.0 / 6.0, 2.0 / 6.0, 2.0 / 6.0, 1.0 / 6.0};
    double x[ORC_NX];
    memcpy(x, x0, sizeof x);
    if (with_sens) {
        memset(S, 0, sizeof(double) * ORC_NX * ORC_NZ);
        for (int i = 0; i < ORC_NX; i++) S[i * ORC_NZ + i] = 1.0;
    }
    for (int m = 0; m < M; m++) {
        double K[ORC_NX], dK[ORC_NX * ORC_NZ], X[ORC_NX], dX[ORC_NX * ORC_NZ];
        double xacc[ORC_NX], Sacc[ORC_NX * ORC_NZ];
        memcpy(xacc, x, sizeof x);
        if (with_sens) memcpy(Sacc, S, sizeof Sacc);
        memset(K, 0, sizeof K);
        memset(dK, 0, sizeof dK);
        for (int st = 0; st < 4; st++) {
            for (int i = 0; i < ORC_NX; i++) X[i] = x[i] + a[st] * h * K[i];
            if (with_sens)
                for (int i = 0; i < ORC_NX * ORC_NZ; i++) dX[i] = S[i] + a[st] * h * dK[i];
            stage_eval(model, with_sens, X, u, s_ref, kappa_ref, nk, dX, K, dK);
            for (int i = 0; i < ORC_NX; i++) xacc[i] += w[st] * h * K[i];
            if (with_sens)
                for (int i = 0; i < ORC_NX * ORC_NZ; i++) Sacc[i] += w[st] * h * dK[i];
        }
        memcpy(x, xacc, sizeof x);
        if (with_sens) memcpy(S, Sacc, sizeof Sacc);
    }
    memcpy(xn, x, sizeof x);
}

void orc_rk4(int model, int integrator, const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots, double dt, int M, double *xnext)
{
    rk4_core(model, integrator, 0, x, u, s_ref, kappa_ref, nknots, dt, M, xnext, 0);
}

void orc_rk4_sens(int model, int integrator, const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots, double dt, int M,
                  double *xnext, double *A, double *Bm)
{
    double S[ORC_NX * ORC_NZ];
    rk4_core(model, integrator, 1, x, u, s_ref, kappa_ref, nknots, dt, M, xnext, S);
    for (int i = 0; i < ORC_NX; i++) {
        for (int j = 0; j < ORC_NX; j++) A[i * ORC_NX + j] = S[i * ORC_NZ + j];
        for (int j = 0; j < ORC_NU; j++) Bm[i * ORC_NU + j] = S[i * ORC_NZ + ORC_NX + j];
    }
}
