/*
 * ihm2_oracle_dyn10.c -- CPU ORACLE (test infrastructure, NOT the product).
 * The 15-state Frenet model with wheel speeds, `fdyn10` (the DYN10 plant of the reference's MiL loop).
 *
 * Follows (reference file:line, relative to /root/reference):
 *   fdyn10_model      python/models.py:609-801 (implicit residual: solved here for xdot -- the normal loads are affine in
 *                     (a_x, a_y), the tyre forces are normal load x Pacejka coefficient, so m a = F is a 2x2 linear system)
 *   lon / lat Pacejka python/models.py:69-80, constants python/constants.py:43-111
 *   call site         python/main.py:490-502 (plant under the Stanley controller)
 * State  x = (s, n, psi, v_x, v_y, r, omega_FL, omega_FR, omega_RL, omega_RR, tau_FL, tau_FR, tau_RL, tau_RR, delta),
 * input  u = (u_tau_FL, u_tau_FR, u_tau_RL, u_tau_RR, u_delta).
 * Integrator: classical RK4 x M (the reference integrates its plants with Radau IIA x 100; this model is singular at standstill,
 * smooth_abs_nonzero(0) = 1e-6 in the slip-ratio denominator, so RK4 is for moving cars only -- DESIGN.md section 8).
 */
#include "ihm2_oracle.h"

#include <math.h>
#include <string.h>

static const double g_ = 9.81, m_ = 230.0, I_z = 137.583, z_CG = 0.295, front_track = 1.24, rear_track = 1.24;
static const double l_R = 0.7853, l_F = 0.7853, wheelbase = 1.5706;
static const double C_r0 = 297.030, C_r1 = 16.665, C_r2 = 0.6784, C_downforce = 3.96864;
static const double b1s = -6.75e-6, b2s = 1.35e-1, b3s = 1.2e-3, c1s = 1.86, d1s = 1.12e-4, d2s = 1.57, e1s = -5.38e-6, e2s = 1.11e-2, e3s = -4.26;
static const double b1a = 3.79e1, b2a = 5.28e2, c1a = 1.57, d1a = -2.03e-4, d2a = 1.77, e1a = -2.24e-3, e2a = 1.81;
static const double R_w = 0.20809, I_w = 0.3, k_d = 0.17, k_s = 15.0, t_T = 1e-3, t_delta = 0.02;

static double sabs_nz(double v) { return tanh(10.0 * v) * v + 1e-6 * exp(-v * v); }

void orc_f_dyn10(const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots, double *xdot)
{
    const double s = x[0], n = x[1], psi = x[2], v_x = x[3], v_y = x[4], r = x[5], delta = x[14];
    const double W0 = 0.5 * m_ * g_ * l_F / wheelbase;
    const double BCDs = (b1s * W0 * W0 + b2s * W0) * exp(-b3s * W0), Cs = c1s, Ds = d1s * W0 + d2s, Es = e1s * W0 * W0 + e2s * W0 + e3s, Bs = BCDs / (Cs * Ds);
    const double BCDa = b1a * sin(2.0 * atan(W0 / b2a)), Ca = c1a, Da = d1a * W0 + d2a, Ea = e1a * W0 + e2a, Ba = BCDa / (Ca * Da);
    const double sd = sin(delta), cd = cos(delta);
    const double F_drag = -(C_r0 + C_r1 * v_x + C_r2 * v_x * v_x) * tanh(1000.0 * v_x);
    const double base = W0 + 0.25 * (0.5 * C_downforce * v_x * v_x);
    const double cx = 0.5 * m_ * z_CG / wheelbase, cy = 0.5 * m_ * z_CG / front_track;
    /* wheel order FL, FR, RL, RR */
    const double vxF[2] = {v_x - 0.5 * front_track * r, v_x + 0.5 * front_track * r}, vyF = v_y + l_F * r;
    double v_lon[4], v_lat[4];
    for (int w = 0; w < 2; w++) { v_lon[w] = cd * vxF[w] + sd * vyF; v_lat[w] = -sd * vxF[w] + cd * vyF; }
    v_lon[2] = v_x - 0.5 * rear_track * r; v_lon[3] = v_x + 0.5 * rear_track * r; v_lat[2] = v_lat[3] = v_y - l_R * r;
    double cl[4], cs[4], fx[4], fy[4];
    for (int w = 0; w < 4; w++) {
        const double va = sabs_nz(v_lon[w]);
        const double alpha = atan2(v_lat[w], va), sr = x[6 + w] * R_w / va - 1.0;
        const double Ba_a = Ba * alpha, Bs_s = Bs * sr;
        cl[w] = Da * sin(Ca * atan(Ba_a - Ea * (Ba_a - atan(Ba_a))));
        cs[w] = Ds * sin(Cs * atan(Bs_s - Es * (Bs_s - atan(Bs_s))));
        /* body-frame force per unit of normal load N_w = -F_z,w:  F_lon = N cs, F_lat = -N cl */
        if (w < 2) { fx[w] = cd * cs[w] + sd * cl[w]; fy[w] = sd * cs[w] - cd * cl[w]; }
        else { fx[w] = cs[w]; fy[w] = -cl[w]; }
    }
    /* N_w = base + sx_w cx a_x + sy_w cy a_y */
    static const double sx[4] = {-1, -1, 1, 1}, sy[4] = {1, -1, 1, -1};
    double Sfx = 0, Sfy = 0, Sxx = 0, Sxy = 0, Syx = 0, Syy = 0;
    for (int w = 0; w < 4; w++) { Sfx += fx[w]; Sfy += fy[w]; Sxx += sx[w] * fx[w]; Sxy += sy[w] * fx[w]; Syx += sx[w] * fy[w]; Syy += sy[w] * fy[w]; }
    const double a11 = m_ - cx * Sxx, a12 = -cy * Sxy, a21 = -cx * Syx, a22 = m_ - cy * Syy;
    const double b1 = F_drag + base * Sfx, b2 = base * Sfy, det = a11 * a22 - a12 * a21;
    const double a_x = (b1 * a22 - a12 * b2) / det, a_y = (a11 * b2 - a21 * b1) / det;
    double Fx[4], Fy[4], Flon[4];
    for (int w = 0; w < 4; w++) {
        const double Nw = base + sx[w] * cx * a_x + sy[w] * cy * a_y;
        Fx[w] = Nw * fx[w]; Fy[w] = Nw * fy[w]; Flon[w] = Nw * cs[w];
    }
    double dk;
    const double kap = orc_kappa(s_ref, kappa_ref, nknots, s, &dk);
    const double s_dot = (v_x * cos(psi) - v_y * sin(psi)) / (1.0 + kap * n);
    xdot[0] = s_dot;
    xdot[1] = v_x * sin(psi) + v_y * cos(psi);
    xdot[2] = r - kap * s_dot;
    xdot[3] = a_x + v_y * r;
    xdot[4] = a_y - v_x * r;
    xdot[5] = ((Fx[1] - Fx[0]) * 0.5 * front_track + (Fy[1] + Fy[0]) * l_F + (Fx[3] - Fx[2]) * 0.5 * rear_track - (Fy[3] + Fy[2]) * l_R) / I_z;
    for (int w = 0; w < 4; w++) {
        xdot[6 + w] = (x[10 + w] - (k_d * x[6 + w] + k_s + R_w * Flon[w])) / I_w;
        xdot[10 + w] = (u[w] - x[10 + w]) / t_T;
    }
    xdot[14] = (u[4] - delta) / t_delta;
}

/* plant step: RK4 x M over dt (tableau dpc/main.py:87-97) */
void orc_sim_step_dyn10(int B, int M, double dt, const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots,
                        double *xnext)
{
    const double h = dt / M;
    for (int b = 0; b < B; b++) {
        double xc[15], k1[15], k2[15], k3[15], k4[15], xt[15];
        const double *ub = u + (size_t)b * 5;
        memcpy(xc, x + (size_t)b * 15, sizeof xc);
        for (int mm = 0; mm < M; mm++) {
            orc_f_dyn10(xc, ub, s_ref, kappa_ref, nknots, k1);
            for (int i = 0; i < 15; i++) xt[i] = xc[i] + 0.5 * h * k1[i];
            orc_f_dyn10(xt, ub, s_ref, kappa_ref, nknots, k2);
            for (int i = 0; i < 15; i++) xt[i] = xc[i] + 0.5 * h * k2[i];
            orc_f_dyn10(xt, ub, s_ref, kappa_ref, nknots, k3);
            for (int i = 0; i < 15; i++) xt[i] = xc[i] + h * k3[i];
            orc_f_dyn10(xt, ub, s_ref, kappa_ref, nknots, k4);
            for (int i = 0; i < 15; i++) xc[i] += h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        }
        memcpy(xnext + (size_t)b * 15, xc, sizeof xc);
    }
}
