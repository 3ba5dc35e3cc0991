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
 * Integrators: (i) the reference's: IRK, 4 Radau IIA stages, 100 steps over dt (python/main.py:395-400, python/sim.py:28-33) --
 * orc_sim_step_dyn10_irk below, usable from rest (python/main.py:438-441 starts at v = 0, where smooth_abs_nonzero(0) = 1e-6 in the
 * slip-ratio denominators makes the wheel-slip dynamics stiff beyond any explicit method); (ii) classical RK4 x M, for moving cars only.
 */
#include "ihm2_oracle.h"

#include <complex.h>
#include <math.h>
#include <string.h>

#include "irk_tableaux.h"

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


/* ---- the same model over dual numbers (C99 complex with first-order-exact elementary functions, as ihm2_oracle_model.c):
 * value = real part, derivative along the seeded direction = imaginary part; the Jacobian of the collocation step's Newton iteration ---- */
typedef double complex cplx;
static inline cplx cs_make(double a, double b) { return a + b * I; }
static inline cplx cs_sin(cplx z) { double a = creal(z); return cs_make(sin(a), cimag(z) * cos(a)); }
static inline cplx cs_cos(cplx z) { double a = creal(z); return cs_make(cos(a), -cimag(z) * sin(a)); }
static inline cplx cs_atan(cplx z) { double a = creal(z); return cs_make(atan(a), cimag(z) / (1.0 + a * a)); }
static inline cplx cs_tanh(cplx z) { double a = creal(z), t = tanh(a); return cs_make(t, cimag(z) * (1.0 - t * t)); }
static inline cplx cs_exp(cplx z) { double e = exp(creal(z)); return cs_make(e, cimag(z) * e); }
/* products and quotients written out: C's complex multiplication would add the second-order term -b1 b2 to the value */
static inline cplx cs_mul(cplx a, cplx b) { return cs_make(creal(a) * creal(b), creal(a) * cimag(b) + cimag(a) * creal(b)); }
static inline cplx cs_div(cplx a, cplx b) { double q = creal(a) / creal(b); return cs_make(q, (cimag(a) - q * cimag(b)) / creal(b)); }
static inline cplx cs_sabs_nz(cplx v) { return cs_mul(cs_tanh(10.0 * v), v) + 1e-6 * cs_exp(-cs_mul(v, v)); }
/* atan2(y, x) with x > 0 (x = smooth_abs_nonzero(.) > 0 at the call sites) */
static inline cplx cs_atan2_pos(cplx y, cplx x) { double yr = creal(y), xr = creal(x); return cs_make(atan2(yr, xr), (xr * cimag(y) - yr * cimag(x)) / (xr * xr + yr * yr)); }

static void f_dyn10_dual(const cplx *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots, cplx *xdot)
{
    const cplx s = x[0], n = x[1], psi = x[2], v_x = x[3], v_y = x[4], r = x[5], delta = x[14];
    const double W0 = 0.5 * m_ * g_ * l_F / wheelbase;
    const double BCDs = (b1s * W0 * W0 + b2s * W0) * exp(-b3s * W0), Cs = c1s, Ds = d1s * W0 + d2s, Es = e1s * W0 * W0 + e2s * W0 + e3s, Bs = BCDs / (Cs * Ds);
    const double BCDa = b1a * sin(2.0 * atan(W0 / b2a)), Ca = c1a, Da = d1a * W0 + d2a, Ea = e1a * W0 + e2a, Ba = BCDa / (Ca * Da);
    const cplx sd = cs_sin(delta), cd = cs_cos(delta);
    const cplx F_drag = -cs_mul(C_r0 + C_r1 * v_x + C_r2 * cs_mul(v_x, v_x), cs_tanh(1000.0 * v_x));
    const cplx base = W0 + 0.25 * (0.5 * C_downforce * cs_mul(v_x, v_x));
    const double cx = 0.5 * m_ * z_CG / wheelbase, cy = 0.5 * m_ * z_CG / front_track;
    const cplx vxF[2] = {v_x - 0.5 * front_track * r, v_x + 0.5 * front_track * r}, vyF = v_y + l_F * r;
    cplx v_lon[4], v_lat[4];
    for (int w = 0; w < 2; w++) { v_lon[w] = cs_mul(cd, vxF[w]) + cs_mul(sd, vyF); v_lat[w] = -cs_mul(sd, vxF[w]) + cs_mul(cd, vyF); }
    v_lon[2] = v_x - 0.5 * rear_track * r; v_lon[3] = v_x + 0.5 * rear_track * r; v_lat[2] = v_lat[3] = v_y - l_R * r;
    cplx cl[4], cs[4], fx[4], fy[4];
    for (int w = 0; w < 4; w++) {
        const cplx va = cs_sabs_nz(v_lon[w]);
        const cplx alpha = cs_atan2_pos(v_lat[w], va), sr = cs_div(x[6 + w] * R_w, va) - 1.0;
        const cplx Ba_a = Ba * alpha, Bs_s = Bs * sr;
        cl[w] = Da * cs_sin(Ca * cs_atan(Ba_a - Ea * (Ba_a - cs_atan(Ba_a))));
        cs[w] = Ds * cs_sin(Cs * cs_atan(Bs_s - Es * (Bs_s - cs_atan(Bs_s))));
        if (w < 2) { fx[w] = cs_mul(cd, cs[w]) + cs_mul(sd, cl[w]); fy[w] = cs_mul(sd, cs[w]) - cs_mul(cd, cl[w]); }
        else { fx[w] = cs[w]; fy[w] = -cl[w]; }
    }
    static const double sx[4] = {-1, -1, 1, 1}, sy[4] = {1, -1, 1, -1};
    cplx Sfx = 0, Sfy = 0, Sxx = 0, Sxy = 0, Syx = 0, Syy = 0;
    for (int w = 0; w < 4; w++) { Sfx += fx[w]; Sfy += fy[w]; Sxx += sx[w] * fx[w]; Sxy += sy[w] * fx[w]; Syx += sx[w] * fy[w]; Syy += sy[w] * fy[w]; }
    const cplx a11 = m_ - cx * Sxx, a12 = -cy * Sxy, a21 = -cx * Syx, a22 = m_ - cy * Syy;
    const cplx b1 = F_drag + cs_mul(base, Sfx), b2 = cs_mul(base, Sfy), det = cs_mul(a11, a22) - cs_mul(a12, a21);
    const cplx a_x = cs_div(cs_mul(b1, a22) - cs_mul(a12, b2), det), a_y = cs_div(cs_mul(a11, b2) - cs_mul(a21, b1), det);
    cplx Fx[4], Fy[4], Flon[4];
    for (int w = 0; w < 4; w++) {
        const cplx Nw = base + sx[w] * cx * a_x + sy[w] * cy * a_y;
        Fx[w] = cs_mul(Nw, fx[w]); Fy[w] = cs_mul(Nw, fy[w]); Flon[w] = cs_mul(Nw, cs[w]);
    }
    double dk;
    const double kr = orc_kappa(s_ref, kappa_ref, nknots, creal(s), &dk);
    const cplx kap = cs_make(kr, cimag(s) * dk);
    const cplx s_dot = cs_div(cs_mul(v_x, cs_cos(psi)) - cs_mul(v_y, cs_sin(psi)), 1.0 + cs_mul(kap, n));
    xdot[0] = s_dot;
    xdot[1] = cs_mul(v_x, cs_sin(psi)) + cs_mul(v_y, cs_cos(psi));
    xdot[2] = r - cs_mul(kap, s_dot);
    xdot[3] = a_x + cs_mul(v_y, r);
    xdot[4] = a_y - cs_mul(v_x, r);
    xdot[5] = ((Fx[1] - Fx[0]) * 0.5 * front_track + (Fy[1] + Fy[0]) * l_F + (Fx[3] - Fx[2]) * 0.5 * rear_track - (Fy[3] + Fy[2]) * l_R) / I_z;
    for (int w = 0; w < 4; w++) {
        xdot[6 + w] = (x[10 + w] - (k_d * x[6 + w] + k_s + R_w * Flon[w])) / I_w;
        xdot[10 + w] = (u[w] - x[10 + w]) / t_T;
    }
    xdot[14] = (u[4] - delta) / t_delta;
}

/* f (15) and J = d f / d x (15 x 15, row-major): one dual evaluation per state direction */
void orc_jac_dyn10(const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots, double *f, double *J)
{
    cplx xc[15], fc[15];
    for (int j = 0; j < 15; j++) {
        for (int i = 0; i < 15; i++) xc[i] = cs_make(x[i], (i == j) ? 1.0 : 0.0);
        f_dyn10_dual(xc, u, s_ref, kappa_ref, nknots, fc);
        for (int i = 0; i < 15; i++) J[i * 15 + j] = cimag(fc[i]);
        if (j == 0) for (int i = 0; i < 15; i++) f[i] = creal(fc[i]);
    }
}

/* dense LU with partial pivoting, in place (as ihm2_oracle_model.c); returns 0 if singular */
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
static void lu_solve(int n, const double *Mx, const int *piv, double *rhs)
{
    for (int c = 0; c < n; c++) if (piv[c] != c) { double t = rhs[c]; rhs[c] = rhs[piv[c]]; rhs[piv[c]] = t; }
    for (int c = 0; c < n; c++) for (int r = c + 1; r < n; r++) rhs[r] -= Mx[r * n + c] * rhs[c];
    for (int c = n - 1; c >= 0; c--) {
        rhs[c] /= Mx[c * n + c];
        for (int r = 0; r < c; r++) rhs[r] -= Mx[r * n + c] * rhs[c];
    }
}

/* Plant step with the reference's integrator (python/main.py:395-400: IRK, GAUSS_RADAU_IIA, 4 stages, 100 steps over dt), made robust
 * enough to start from rest (python/main.py:438-441).  acados runs a FIXED number of Newton iterations per step (3) from K = 0; on
 * this model at v = 0 that iteration does not converge (the slip ratio omega R_w / smooth_abs_nonzero(v) - 1 has a basin of 1e-7 rad/s
 * around the rolling condition; tests/test_oracle_dyn10.py shows 3, 6 and 12 iterations giving three different answers), so the
 * collocation equations K_i = f(x + h sum_j A_ij K_j, u) are solved here TO CONVERGENCE, with the step length driven by the Newton
 * iteration as in any production implicit integrator:
 *   - predictor: K_i = the derivative at the end of the previous step (f(x) at the very first one);
 *   - Newton with a fresh Jacobian I - h (A (x) J_i) per iteration (60 x 60, dense LU with partial pivoting), at most newton_iter
 *     iterations; converged when max_i |dK_i| / (1 + |K_i|) <= 1e-10;
 *   - not converged (or NaN): the step is retried with h / 4; converged in at most 4 iterations: the next step tries 2 h; h never
 *     exceeds dt / M (the reference's grid) and steps are clipped to end exactly at dt.
 * Away from standstill every step converges in 2-3 iterations and the grid is the reference's 100 equal steps. */
void orc_sim_step_dyn10_irk(int B, int integrator, int M, int newton_iter, double dt, const double *x, const double *u, const double *s_ref,
                            const double *kappa_ref, int nknots, double *xnext)
{
    const double (*At)[4] = (integrator == ORC_INTEG_IRK_GL4) ? IRK_GL4_A : IRK_RADAU4_A;
    const double *bt = (integrator == ORC_INTEG_IRK_GL4) ? IRK_GL4_b : IRK_RADAU4_b;
    enum { NXD = 15, NK = 60 };
    const double h_max = dt / M, h_min = dt * 1e-12;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < B; b++) {
        double xc[NXD], K[4][NXD], F[4][NXD], J[4][NXD * NXD], Mx[NK * NK], rr[NK], kend[NXD];
        int piv[NK];
        const double *ub = u + (size_t)b * 5;
        memcpy(xc, x + (size_t)b * NXD, sizeof xc);
        orc_f_dyn10(xc, ub, s_ref, kappa_ref, nknots, kend);
        double t = 0.0, h = h_max;
        int failed = 0;
        while (t < dt * (1.0 - 1e-14) && !failed) {
            if (h > dt - t) h = dt - t;
            for (int i = 0; i < 4; i++) memcpy(K[i], kend, sizeof kend);
            int conv = 0, it;
            for (it = 0; it < newton_iter && !conv; it++) {
                for (int i = 0; i < 4; i++) {
                    double X[NXD];
                    for (int a = 0; a < NXD; a++) {
                        double acc = xc[a];
                        for (int j = 0; j < 4; j++) acc += h * At[i][j] * K[j][a];
                        X[a] = acc;
                    }
                    orc_jac_dyn10(X, ub, s_ref, kappa_ref, nknots, F[i], J[i]);
                }
                for (int i = 0; i < 4; i++)
                    for (int a = 0; a < NXD; a++)
                        for (int j = 0; j < 4; j++)
                            for (int c = 0; c < NXD; c++)
                                Mx[(i * NXD + a) * NK + j * NXD + c] = ((i == j && a == c) ? 1.0 : 0.0) - h * At[i][j] * J[i][a * NXD + c];
                if (!lu_factor(NK, Mx, piv)) break;
                for (int i = 0; i < 4; i++) for (int a = 0; a < NXD; a++) rr[i * NXD + a] = -(K[i][a] - F[i][a]);
                lu_solve(NK, Mx, piv, rr);
                double dmax = 0.0;
                for (int i = 0; i < 4; i++)
                    for (int a = 0; a < NXD; a++) {
                        K[i][a] += rr[i * NXD + a];
                        const double d = fabs(rr[i * NXD + a]) / (1.0 + fabs(K[i][a]));
                        dmax = (d == d) ? fmax(dmax, d) : INFINITY;
                    }
                if (!(dmax < INFINITY)) break;
                conv = dmax <= 1e-10;
            }
            if (!conv) {
                h *= 0.25;
                if (h < h_min) failed = 1;
                continue;
            }
            for (int a = 0; a < NXD; a++) {
                double acc = xc[a];
                for (int i = 0; i < 4; i++) acc += h * bt[i] * K[i][a];
                xc[a] = acc;
            }
            /* derivative at the end of the step: the last Radau IIA stage sits there (c_4 = 1); Gauss-Legendre re-evaluates */
            if (integrator == ORC_INTEG_IRK_GL4) orc_f_dyn10(xc, ub, s_ref, kappa_ref, nknots, kend);
            else memcpy(kend, K[3], sizeof kend);
            t += h;
            if (it <= 4) h = fmin(2.0 * h, h_max);
        }
        for (int a = 0; a < NXD; a++) xnext[(size_t)b * NXD + a] = failed ? NAN : xc[a];
    }
}
