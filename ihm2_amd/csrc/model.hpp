// model.hpp -- device-side vehicle models of the ihm2 NMPC (double precision).
//
// fkin6: Frenet kinematic 6-DOF bicycle with first-order actuators, reference python/models.py:232-307.
// fdyn6: Frenet 4-wheel Pacejka model, reference python/models.py:455-606 (implicit there; solved
//        here for xdot through the 2x2 system in (a_x, a_y)).
// Car parameters: reference python/constants.py:43-111.
//
// The Jacobian of fkin6 is hand-derived and stored sparsely: 31 of 80 entries are structurally
// non-zero and the state splits as (T,delta) -> (v_x,v_y,r) -> (s,n,psi) (block triangular).
#pragma once

#include <hip/hip_runtime.h>

namespace ihm2 {

// ---- constants (python/constants.py) ----
constexpr double k_g = 9.81;
constexpr double k_m = 230.0;
constexpr double k_Iz = 137.583;
constexpr double k_zCG = 0.295;
constexpr double k_axle_track = 1.24;
constexpr double k_lR = 0.7853;
constexpr double k_lF = 0.7853;
constexpr double k_wheelbase = 1.5706;
constexpr double k_rwd = k_lR / k_wheelbase;   // rear_weight_distribution
constexpr double k_Cm0 = 4.950;
constexpr double k_Cr0 = 297.030;
constexpr double k_Cr1 = 16.665;
constexpr double k_Cr2 = 0.6784;
constexpr double k_tT = 1e-3;
constexpr double k_tdelta = 0.02;
constexpr double k_Cdown = 3.96864;
constexpr double k_Ktv = 300.0;
// lateral Pacejka, constant-load version (python/constants.py:84-95): values of Ba, Ca, Da, Ea
constexpr double k_static_weight = 0.5 * k_m * k_g * k_lF / k_wheelbase;
constexpr double k_b1a = 3.79e1, k_b2a = 5.28e2, k_c1a = 1.57, k_d1a = -2.03e-4, k_d2a = 1.77, k_e1a = -2.24e-3, k_e2a = 1.81;

// ---- curvature table: piecewise-linear kappa(s) with a carried segment ----
// The lookup is "exact" (binary search) once, then hunts from the previous segment: along an RK4
// trajectory s moves by centimetres per stage while the knots are ~0.68 m apart.
struct TrackSeg {
    const double *s_ref;
    const double *k_ref;
    int n;
    int idx;
    double s_lo, s_hi, k_lo, slope;

    __device__ __forceinline__ void load(int i) {
        idx = i;
        s_lo = s_ref[i];
        s_hi = s_ref[i + 1];
        k_lo = k_ref[i];
        slope = (k_ref[i + 1] - k_lo) / (s_hi - s_lo);
    }
    __device__ void init(const double *sr, const double *kr, int nknots, double s) {
        s_ref = sr; k_ref = kr; n = nknots;
        int lo = 0, hi = n - 1;
        if (!(s >= sr[0])) { lo = 0; }
        else if (s >= sr[n - 1]) { lo = n - 2; }
        else {
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (sr[mid] <= s) lo = mid; else hi = mid;
            }
        }
        load(lo);
    }
    // largest i in [0, n-2] with s_ref[i] <= s; linear extrapolation outside the table
    __device__ __forceinline__ void seek(double s) {
        // bounded walks: a NaN s fails both comparisons and leaves the segment unchanged
        while (s >= s_hi && idx < n - 2) load(idx + 1);
        while (s < s_lo && idx > 0) load(idx - 1);
    }
    __device__ __forceinline__ double kappa(double s, double &dk) {
        seek(s);
        dk = slope;
        return k_lo + slope * (s - s_lo);
    }
};

// ---- fkin6: xdot and the 31 structural non-zeros of d xdot / d (x,u) ----
// J[i][j], j < 8: d/dx_j ; j = 8: d/du_T ; j = 9: d/du_delta.  Entries that are structurally zero
// are never written and never read.
template <bool WITH_JAC>
__device__ __forceinline__ void fkin6_eval(const double (&x)[8], double u_T, double u_delta, TrackSeg &trk,
                                           double (&f)[8], double (&J)[8][10])
{
    const double c = k_rwd;
    const double n = x[1], psi = x[2], v_x = x[3], v_y = x[4], r = x[5], T = x[6], delta = x[7];
    const double delta_dot = (u_delta - delta) * (1.0 / k_tdelta);
    const double T_dot = (u_T - T) * (1.0 / k_tT);
    // longitudinal forces (models.py:255-258)
    const double F_motor = k_Cm0 * T;
    const double sg = tanh(10.0 * v_x);
    const double poly = k_Cr0 + k_Cr1 * v_x + k_Cr2 * v_x * v_x;
    const double F_drag = -poly * sg;
    const double F_Rx = 0.5 * F_motor + F_drag, F_Fx = 0.5 * F_motor;
    // slip angle of the kinematic model (models.py:261-284)
    const double td = tan(delta);
    const double beta = atan(c * td);
    double sb, cb;
    sincos(beta, &sb, &cb);
    const double den = 1.0 + c * c * td * td;
    const double bp = c * (1.0 + td * td) / den;          // d beta / d delta
    const double beta_dot = bp * delta_dot;
    double sdb, cdb;
    sincos(delta - beta, &sdb, &cdb);
    const double v_dot = (F_Rx * cb + F_Fx * cdb) * (1.0 / k_m);
    // Frenet kinematics (models.py:290-301)
    double dk;
    const double kap = trk.kappa(x[0], dk);
    double sp, cp;
    sincos(psi, &sp, &cp);
    const double num = v_x * cp - v_y * sp;
    const double dn = 1.0 + kap * n;
    const double inv_dn = 1.0 / dn;
    const double s_dot = num * inv_dn;
    const double v_y_dot = v_dot * sb + beta_dot * v_x;
    f[0] = s_dot;
    f[1] = v_x * sp + v_y * cp;
    f[2] = r - kap * s_dot;
    f[3] = v_dot * cb - beta_dot * v_y;
    f[4] = v_y_dot;
    f[5] = k_lR * v_y_dot - beta_dot;
    f[6] = T_dot;
    f[7] = delta_dot;
    if (WITH_JAC) {
        const double dFdrag = -(k_Cr1 + 2.0 * k_Cr2 * v_x) * sg - poly * 10.0 * (1.0 - sg * sg);
        const double bpp = 2.0 * c * td * (1.0 + td * td) * (1.0 - c * c) / (den * den);
        const double dbd_dd = bpp * delta_dot - bp * (1.0 / k_tdelta);
        const double dbd_du = bp * (1.0 / k_tdelta);
        const double dv_dvx = dFdrag * cb * (1.0 / k_m);
        const double dv_dT = 0.5 * k_Cm0 * (cb + cdb) * (1.0 / k_m);
        const double dv_dd = (-F_Rx * sb * bp - F_Fx * sdb * (1.0 - bp)) * (1.0 / k_m);
        const double q = -s_dot * inv_dn;          // d s_dot / d (kappa n)
        J[0][0] = q * dk * n;
        J[0][1] = q * kap;
        J[0][2] = (-v_x * sp - v_y * cp) * inv_dn;
        J[0][3] = cp * inv_dn;
        J[0][4] = -sp * inv_dn;
        J[1][2] = num;
        J[1][3] = sp;
        J[1][4] = cp;
        J[2][0] = -dk * s_dot - kap * J[0][0];
        J[2][1] = -kap * J[0][1];
        J[2][2] = -kap * J[0][2];
        J[2][3] = -kap * J[0][3];
        J[2][4] = -kap * J[0][4];
        J[2][5] = 1.0;
        J[3][3] = dv_dvx * cb;
        J[3][4] = -beta_dot;
        J[3][6] = dv_dT * cb;
        J[3][7] = dv_dd * cb - v_dot * sb * bp - dbd_dd * v_y;
        J[3][9] = -dbd_du * v_y;
        J[4][3] = dv_dvx * sb + beta_dot;
        J[4][6] = dv_dT * sb;
        J[4][7] = dv_dd * sb + v_dot * cb * bp + dbd_dd * v_x;
        J[4][9] = dbd_du * v_x;
        J[5][3] = k_lR * J[4][3];
        J[5][6] = k_lR * J[4][6];
        J[5][7] = k_lR * J[4][7] - dbd_dd;
        J[5][9] = k_lR * J[4][9] - dbd_du;
        J[6][6] = -1.0 / k_tT;
        J[6][8] = 1.0 / k_tT;
        J[7][7] = -1.0 / k_tdelta;
        J[7][9] = 1.0 / k_tdelta;
    }
}

// Structural pattern of d f_i / d x_l (bit l of JX_MASK[i]) and of d f_i / d u (bit 0: u_T, bit 1: u_delta)
__device__ constexpr unsigned JX_MASK[8] = {0x1Fu, 0x1Cu, 0x3Fu, 0xD8u, 0xC8u, 0xC8u, 0x40u, 0x80u};
__device__ constexpr unsigned JU_MASK[8] = {0u, 0u, 0u, 2u, 2u, 2u, 1u, 2u};
// Rows of the sensitivity matrix S = d x_m / d (x_0, u) that can be non-zero in column j
// (block-triangular structure): columns s0,n0,psi0 | v_x0,v_y0 | r0 | T0 | delta0 | u_T | u_delta
__device__ constexpr unsigned S_COL_MASK[10] = {0x07u, 0x07u, 0x07u, 0x3Fu, 0x3Fu, 0x27u, 0x7Fu, 0xBFu, 0x7Fu, 0xBFu};

// ---- fdyn6 (plant): explicit xdot ----
__device__ __forceinline__ double lat_pacejka(double alpha) {
    const double BCDa = k_b1a * sin(2.0 * atan(k_static_weight / k_b2a));
    const double Ca = k_c1a, Da = k_d1a * k_static_weight + k_d2a, Ea = k_e1a * k_static_weight + k_e2a;
    const double Ba = BCDa / (Ca * Da);
    const double Bx = Ba * alpha;
    return Da * sin(Ca * atan(Bx - Ea * (Bx - atan(Bx))));
}
__device__ __forceinline__ double smooth_abs_nonzero(double v) { return tanh(10.0 * v) * v + 1e-6 * exp(-v * v); }

__device__ inline void fdyn6_eval(const double (&x)[8], double u_T, double u_delta, TrackSeg &trk, double (&f)[8])
{
    const double n = x[1], psi = x[2], v_x = x[3], v_y = x[4], r = x[5], T = x[6], delta = x[7];
    double sd, cd;
    sincos(delta, &sd, &cd);
    const double F_down = 0.5 * k_Cdown * v_x * v_x;
    const double cx = 0.5 * k_m * k_zCG / k_wheelbase, cy = 0.5 * k_m * k_zCG / k_axle_track;
    const double base = k_static_weight + 0.25 * F_down;
    const double sxv[4] = {-1.0, -1.0, 1.0, 1.0}, syv[4] = {1.0, -1.0, 1.0, -1.0};   // FL FR RL RR
    const double hx = 0.5 * k_axle_track;
    const double v_x_FL = v_x - hx * r, v_x_FR = v_x + hx * r, v_y_F = v_y + k_lF * r;
    const double v_lon_FL = cd * v_x_FL + sd * v_y_F, v_lon_FR = cd * v_x_FR + sd * v_y_F;
    const double v_lat_FL = -sd * v_x_FL + cd * v_y_F, v_lat_FR = -sd * v_x_FR + cd * v_y_F;
    const double v_lon_RL = v_x - hx * r, v_lon_RR = v_x + hx * r, v_lat_R = v_y - k_lR * r;
    const double a_FL = atan2(v_lat_FL, smooth_abs_nonzero(v_lon_FL));
    const double a_FR = atan2(v_lat_FR, smooth_abs_nonzero(v_lon_FR));
    const double a_RL = atan2(v_lat_R, smooth_abs_nonzero(v_lon_RL));
    const double a_RR = atan2(v_lat_R, smooth_abs_nonzero(v_lon_RR));
    // crossed slip angles exactly as models.py:543-546 (quirk Q3)
    const double glat[4] = {lat_pacejka(a_RR), lat_pacejka(a_RL), lat_pacejka(a_FR), lat_pacejka(a_FL)};
    const double F_drag = -(k_Cr0 + k_Cr1 * v_x + k_Cr2 * v_x * v_x) * tanh(10.0 * v_x);
    const double beta = atan(k_rwd * tan(delta));
    const double r_kin = sqrt(v_x * v_x + v_y * v_y) * sin(beta) / k_lR;
    const double dtau = k_Ktv * (r_kin - r);
    const double denom = -k_m * k_g - 0.25 * F_down;
    const double gm = k_Cm0 * (T - dtau) / denom, gp = k_Cm0 * (T + dtau) / denom;
    const double glon[4] = {gm, gp, gm, gp};
    double cxk[4], cyk[4], czk[4];
    cxk[0] = glon[0] * cd - glat[0] * sd; cyk[0] = glon[0] * sd + glat[0] * cd;
    cxk[1] = glon[1] * cd - glat[1] * sd; cyk[1] = glon[1] * sd + glat[1] * cd;
    cxk[2] = glon[2]; cyk[2] = glat[2];
    cxk[3] = glon[3]; cyk[3] = glat[3];
    czk[0] = -cxk[0] * hx + cyk[0] * k_lF;
    czk[1] = cxk[1] * hx + cyk[1] * k_lF;
    czk[2] = -glon[2] * hx - glat[2] * k_lR;
    czk[3] = glon[3] * hx - glat[3] * k_lR;
    double X0 = F_drag, Xx = 0, Xy = 0, Y0 = 0, Yx = 0, Yy = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        X0 -= cxk[k] * base; Xx -= cxk[k] * (sxv[k] * cx); Xy -= cxk[k] * (syv[k] * cy);
        Y0 -= cyk[k] * base; Yx -= cyk[k] * (sxv[k] * cx); Yy -= cyk[k] * (syv[k] * cy);
    }
    const double a11 = k_m - Xx, a12 = -Xy, a21 = -Yx, a22 = k_m - Yy;
    const double det = a11 * a22 - a12 * a21;
    const double a_x = (X0 * a22 - a12 * Y0) / det;
    const double a_y = (a11 * Y0 - a21 * X0) / det;
    double Mz = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) Mz += czk[k] * -(base + sxv[k] * cx * a_x + syv[k] * cy * a_y);
    double dk;
    const double kap = trk.kappa(x[0], dk);
    double sp, cp;
    sincos(psi, &sp, &cp);
    const double s_dot = (v_x * cp - v_y * sp) / (1.0 + kap * n);
    f[0] = s_dot;
    f[1] = v_x * sp + v_y * cp;
    f[2] = r - kap * s_dot;
    f[3] = a_x + v_y * r;
    f[4] = a_y - v_x * r;
    f[5] = Mz / k_Iz;
    f[6] = (u_T - T) * (1.0 / k_tT);
    f[7] = (u_delta - delta) * (1.0 / k_tdelta);
}

}  // namespace ihm2
