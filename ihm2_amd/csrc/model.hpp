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

// ---- lean sine / cosine / tanh for the model evaluations ----
// The models call sincos on the heading error and the steering angle (|x| << 1) and tanh on 10 v_x, four times per RK4 sub-step and
// interval.  The library versions carry the Payne-Hanek path for huge arguments and ~190 instructions per sincos, ~180 per tanh; these
// take ~50 and ~75: three-constant Cody-Waite reduction (exact for the n = 0 case the models live in, error < 1e-16 |x| up to 1e5),
// fdlibm's kernel polynomials (< 1 ulp on [-pi/4, pi/4]); beyond 1e5 the library function is called.
static __device__ __noinline__ double2 sincos_lib(double x) { double sv, cv; sincos(x, &sv, &cv); return make_double2(sv, cv); }
__device__ __forceinline__ void fast_sincos(double x, double *sp, double *cp)
{
    const double n = rint(x * 6.36619772367581382433e-01);                 // 2 / pi
    double r = fma(-n, 1.57079632679489655800e+00, x);
    r = fma(-n, 6.12323399573676603587e-17, r);
    r = fma(-n, -1.49738490485916983693e-33, r);           // pi/2 = P1 + P2 + P3, to 2^-161
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sv = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double cv = 1.0 - fma(0.5, z, -(z * z) * pc);
    const int q = (int)n;
    double so = (q & 1) ? cv : sv, co = (q & 1) ? sv : cv;
    so = (q & 2) ? -so : so;
    co = ((q + 1) & 2) ? -co : co;
    if (!(fabs(x) <= 1e5)) { const double2 l = sincos_lib(x); so = l.x; co = l.y; }      // huge, inf, nan
    *sp = so; *cp = co;
}
// tanh(y) = sign(y) (1 - t) / (1 + t), t = exp(-2 |y|): absolute error ~1e-16 (the relative error grows towards y = 0, where
// every use multiplies the result by a quantity that vanishes with y)
__device__ __forceinline__ double tanh_e(double y)
{
    const double t = exp(-2.0 * fabs(y));
    return copysign((1.0 - t) / (1.0 + t), y);
}

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
    const double sg = tanh_e(10.0 * v_x);
    const double poly = k_Cr0 + k_Cr1 * v_x + k_Cr2 * v_x * v_x;
    const double F_drag = -poly * sg;
    const double F_Rx = 0.5 * F_motor + F_drag, F_Fx = 0.5 * F_motor;
    // slip angle of the kinematic model (models.py:261-284): beta = atan(c tan(delta)).  One sincos(delta) and
    // one rsqrt replace tan, atan, sincos(beta) and sincos(delta - beta):
    //   cos(beta) = cd / sqrt(cd^2 + c^2 sd^2), sin(beta) = c sd / sqrt(cd^2 + c^2 sd^2)   (cd > 0 for |delta| < pi/2)
    double sd, cd;
    fast_sincos(delta, &sd, &cd);
    const double td = sd / cd;
    const double hyp = 1.0 / sqrt(cd * cd + c * c * sd * sd);
    const double cb = cd * hyp, sb = c * sd * hyp;
    const double den = 1.0 + c * c * td * td;
    const double bp = c * (1.0 + td * td) / den;          // d beta / d delta
    const double beta_dot = bp * delta_dot;
    const double cdb = cd * cb + sd * sb, sdb = sd * cb - cd * sb;      // cos / sin (delta - beta)
    const double v_dot = (F_Rx * cb + F_Fx * cdb) * (1.0 / k_m);
    // Frenet kinematics (models.py:290-301)
    double dk;
    const double kap = trk.kappa(x[0], dk);
    double sp, cp;
    fast_sincos(psi, &sp, &cp);
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

// ---- lateral acceleration of the kinematic model: the fifth nonlinear row of the kinematic constraint set ----
// old/generate_acaods_interface.py:198-209 (`+ ([] if is_dynamic else [a_lat])`), definition :266-271 and old/scripts/gen_mpc.py:182-184:
//   a_lat = (-F_Rx sin(beta) + F_Fx sin(delta - beta)) / m + (v_x^2 + v_y^2) sin(beta) / l_R
// with the forces and the slip angle of fkin6 (python/models.py:255-263).  g = d a_lat / d (v_x, v_y, T, delta): the row's non-zeros.
__device__ __forceinline__ double alat_eval(const double v_x, const double v_y, const double T, const double delta, double (&g)[4])
{
    const double c = k_rwd;
    const double sg = tanh_e(10.0 * v_x);
    const double poly = k_Cr0 + k_Cr1 * v_x + k_Cr2 * v_x * v_x;
    const double F_Rx = 0.5 * k_Cm0 * T - poly * sg, F_Fx = 0.5 * k_Cm0 * T;
    double sd, cd;
    fast_sincos(delta, &sd, &cd);
    const double td = sd / cd;
    const double hyp = 1.0 / sqrt(cd * cd + c * c * sd * sd);
    const double cb = cd * hyp, sb = c * sd * hyp;
    const double bp = c * (1.0 + td * td) / (1.0 + c * c * td * td);      // d beta / d delta
    const double cdb = cd * cb + sd * sb, sdb = sd * cb - cd * sb;          // cos / sin (delta - beta)
    const double vv = v_x * v_x + v_y * v_y;
    const double dFdrag = -(k_Cr1 + 2.0 * k_Cr2 * v_x) * sg - poly * 10.0 * (1.0 - sg * sg);
    g[0] = -dFdrag * sb * (1.0 / k_m) + 2.0 * v_x * sb * (1.0 / k_lR);
    g[1] = 2.0 * v_y * sb * (1.0 / k_lR);
    g[2] = 0.5 * k_Cm0 * (sdb - sb) * (1.0 / k_m);
    g[3] = (-F_Rx * cb * bp + F_Fx * cdb * (1.0 - bp)) * (1.0 / k_m) + vv * cb * bp * (1.0 / k_lR);
    return (-F_Rx * sb + F_Fx * sdb) * (1.0 / k_m) + vv * sb * (1.0 / k_lR);
}

// Structural pattern of d f_i / d x_l (bit l of JX_MASK[model][i]) and of d f_i / d u (bit 0: u_T, bit 1: u_delta).
// fdyn6: rows 3..5 (v_x_dot, v_y_dot, r_dot) depend on (v_x, v_y, r, T, delta) and on no input directly.
__device__ constexpr unsigned JX_MASK[2][8] = {{0x1Fu, 0x1Cu, 0x3Fu, 0xD8u, 0xC8u, 0xC8u, 0x40u, 0x80u},
                                               {0x1Fu, 0x1Cu, 0x3Fu, 0xF8u, 0xF8u, 0xF8u, 0x40u, 0x80u}};
__device__ constexpr unsigned JU_MASK[2][8] = {{0u, 0u, 0u, 2u, 2u, 2u, 1u, 2u}, {0u, 0u, 0u, 0u, 0u, 0u, 1u, 2u}};
// Rows of the sensitivity matrix S = d x_m / d (x_0, u) that can be non-zero in column j
// (block-triangular structure): columns s0,n0,psi0 | v_x0,v_y0 | r0 | T0 | delta0 | u_T | u_delta
__device__ constexpr unsigned S_COL_MASK[2][10] = {{0x07u, 0x07u, 0x07u, 0x3Fu, 0x3Fu, 0x27u, 0x7Fu, 0xBFu, 0x7Fu, 0xBFu},
                                                   {0x07u, 0x07u, 0x07u, 0x3Fu, 0x3Fu, 0x3Fu, 0x7Fu, 0xBFu, 0x7Fu, 0xBFu}};

// ---- transcendental functions of the dynamic model as CALLS ----
// One body per function instead of one inlined copy per use (13 atan, 6 sincos, 5 tanh, 4 exp per evaluation of the force model):
// the linearisation loop shrinks from 61 KB of code to 33 KB, i.e. under the 64 KB instruction cache two CUs share -- measured
// 12.3 -> 6.3 ms for 8192 x 40 intervals, with MORE instructions executed (the inlined copies were specialised per site).
static __device__ __noinline__ double nl_atan(double x) { return atan(x); }
static __device__ __noinline__ double nl_exp(double x) { return exp(x); }
static __device__ __noinline__ double2 nl_sincos(double x) { double sv, cv; fast_sincos(x, &sv, &cv); return make_double2(sv, cv); }
static __device__ __noinline__ double nl_tanh(double y) { return tanh_e(y); }

// The two per-wheel functions are calls as well (four uses each per evaluation), with their own transcendentals inlined: a call
// inside a call costs a link-register spill (7.1 ms with nested calls against 6.3 ms).
// slip angle alpha = atan(v_lat / smooth_abs_nonzero(v_lon)), smooth_abs_nonzero(v) = tanh(10 v) v + 1e-6 exp(-v^2)
// (python/models.py:417-421,533-541), and its two partial derivatives
struct SlipD { double a, d_lat, d_lon; };
static __device__ __noinline__ SlipD nl_slip_d(double v_lat, double v_lon)
{
    const double th = tanh_e(10.0 * v_lon), e = exp(-(v_lon * v_lon));
    const double xs = fma(th, v_lon, 1e-6 * e);
    const double dxs = th + 10.0 * v_lon * fma(-th, th, 1.0) - 2e-6 * v_lon * e;
    const double ix = 1.0 / xs, q = v_lat * ix;
    const double w = ix / fma(q, q, 1.0);            // d atan(q) / dq  x  dq / dv_lat
    SlipD r; r.a = atan(q); r.d_lat = w; r.d_lon = -w * q * dxs;
    return r;
}
static __device__ __noinline__ double nl_slip(double v_lat, double v_lon)
{
    const double th = tanh_e(10.0 * v_lon), e = exp(-(v_lon * v_lon));
    return atan(v_lat / fma(th, v_lon, 1e-6 * e));
}
// lateral Pacejka force coefficient (python/models.py:423-440) and its derivative
struct PacD { double g, dg; };
static __device__ __noinline__ PacD nl_pacejka_d(double alpha)
{
    const double BCDa = k_b1a * sin(2.0 * atan(k_static_weight / k_b2a));
    const double Ca = k_c1a, Da = k_d1a * k_static_weight + k_d2a, Ea = k_e1a * k_static_weight + k_e2a;
    const double Ba = BCDa / (Ca * Da);
    const double Bx = alpha * Ba;
    const double t1 = atan(Bx), dt1 = 1.0 / fma(Bx, Bx, 1.0);
    const double in = Bx - (Bx - t1) * Ea, din = 1.0 - (1.0 - dt1) * Ea;
    const double t2 = atan(in), dt2 = din / fma(in, in, 1.0);
    double sv, cv;
    fast_sincos(t2 * Ca, &sv, &cv);
    PacD r; r.g = sv * Da; r.dg = cv * (Da * Ca * Ba) * dt2;
    return r;
}
static __device__ __noinline__ double nl_pacejka(double alpha)
{
    const double BCDa = k_b1a * sin(2.0 * atan(k_static_weight / k_b2a));
    const double Ca = k_c1a, Da = k_d1a * k_static_weight + k_d2a, Ea = k_e1a * k_static_weight + k_e2a;
    const double Ba = BCDa / (Ca * Da);
    const double Bx = alpha * Ba;
    return sin(atan(Bx - (Bx - atan(Bx)) * Ea) * Ca) * Da;
}

// scalar instantiation of the generic force model (the plants)
__device__ __forceinline__ double msin(double x) { return nl_sincos(x).x; }
__device__ __forceinline__ double matan(double x) { return nl_atan(x); }
__device__ __forceinline__ double mtanh(double x) { return nl_tanh(x); }
__device__ __forceinline__ double msqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double val(double a) { return a; }

// ---- from here to the end of the dynamic model: NO implicit contraction ----
// Left to fuse, the back end turns a*b + c*d into fma(a, b, c*d) or fma(c, d, a*b) by the use counts it happens to see, which depend on what the
// function was inlined into: the same source gave the stand-alone linearisation kernel and the persistent loop torque-direction derivatives one
// rounding apart once scheduling fences and LDS parking changed the surroundings (tests/test_gpu_closed_loop.py compares the two bit for bit).
// The dual-number rules below say fma where they want one (product rule, chain rules); everything else is the operation it is written as, in
// every context.  (This works because the library is compiled with -ffp-contract=on and asks for `fast` by pragma, ihm2mpc_internal.h: the
// command-line `fast` is a global switch of the back end that no pragma turns off.)
#pragma clang fp contract(off)

// ---- sparse forward-mode duals: value + the directional derivatives named by the bits of MASK (bit k = d / d input k) ----
// The force model's intermediates depend on one to five of its inputs (sin(delta): one; a rear slip angle: three; the load
// transfer: all five).  Dual<5> carries five derivative slots through all of them, and without fast-math the compiler may not
// drop the products with the literal zeros of the seeds; SD<MASK> does not have the slots in the first place: the result type of an
// operation is the union of its operands' masks, resolved at compile time.
__host__ __device__ constexpr int sd_pc(unsigned m) { int n = 0; for (; m; m >>= 1) n += (int)(m & 1u); return n; }
__host__ __device__ constexpr int sd_ix(unsigned m, int k) { return sd_pc(m & ((1u << k) - 1u)); }
template <unsigned MASK>
struct SD {
    double v;
    double d[sd_pc(MASK) ? sd_pc(MASK) : 1];
};
#define SD_HAS(M, k) ((((M) >> (k)) & 1u) != 0u)
#define SD_BITS(OP) OP(0) OP(1) OP(2) OP(3) OP(4)
template <unsigned M> __device__ __forceinline__ SD<M> sd_seed(double v)      // the input itself: MASK has exactly one bit
{
    SD<M> r; r.v = v; r.d[0] = 1.0; return r;
}
template <unsigned TO, unsigned M> __device__ __forceinline__ SD<TO> sd_widen(const SD<M> &a)
{
    static_assert((M & ~TO) == 0u, "widening only");
    SD<TO> r; r.v = a.v;
#define OP(k) if constexpr (SD_HAS(TO, k)) { if constexpr (SD_HAS(M, k)) r.d[sd_ix(TO, k)] = a.d[sd_ix(M, k)]; else r.d[sd_ix(TO, k)] = 0.0; }
    SD_BITS(OP)
#undef OP
    return r;
}
template <unsigned A, unsigned B> __device__ __forceinline__ SD<A | B> operator+(const SD<A> &a, const SD<B> &b)
{
    SD<A | B> r; r.v = a.v + b.v;
#define OP(k) if constexpr (SD_HAS(A | B, k)) {                                                                        \
        if constexpr (SD_HAS(A, k) && SD_HAS(B, k)) r.d[sd_ix(A | B, k)] = a.d[sd_ix(A, k)] + b.d[sd_ix(B, k)];         \
        else if constexpr (SD_HAS(A, k)) r.d[sd_ix(A | B, k)] = a.d[sd_ix(A, k)];                                       \
        else r.d[sd_ix(A | B, k)] = b.d[sd_ix(B, k)]; }
    SD_BITS(OP)
#undef OP
    return r;
}
template <unsigned A, unsigned B> __device__ __forceinline__ SD<A | B> operator-(const SD<A> &a, const SD<B> &b)
{
    SD<A | B> r; r.v = a.v - b.v;
#define OP(k) if constexpr (SD_HAS(A | B, k)) {                                                                        \
        if constexpr (SD_HAS(A, k) && SD_HAS(B, k)) r.d[sd_ix(A | B, k)] = a.d[sd_ix(A, k)] - b.d[sd_ix(B, k)];         \
        else if constexpr (SD_HAS(A, k)) r.d[sd_ix(A | B, k)] = a.d[sd_ix(A, k)];                                       \
        else r.d[sd_ix(A | B, k)] = -b.d[sd_ix(B, k)]; }
    SD_BITS(OP)
#undef OP
    return r;
}
template <unsigned A, unsigned B> __device__ __forceinline__ SD<A | B> operator*(const SD<A> &a, const SD<B> &b)
{
    SD<A | B> r; r.v = a.v * b.v;
#define OP(k) if constexpr (SD_HAS(A | B, k)) {                                                                                          \
        if constexpr (SD_HAS(A, k) && SD_HAS(B, k)) r.d[sd_ix(A | B, k)] = fma(a.v, b.d[sd_ix(B, k)], a.d[sd_ix(A, k)] * b.v);            \
        else if constexpr (SD_HAS(A, k)) r.d[sd_ix(A | B, k)] = a.d[sd_ix(A, k)] * b.v;                                                   \
        else r.d[sd_ix(A | B, k)] = a.v * b.d[sd_ix(B, k)]; }
    SD_BITS(OP)
#undef OP
    return r;
}
template <unsigned A, unsigned B> __device__ __forceinline__ SD<A | B> operator/(const SD<A> &a, const SD<B> &b)
{
    SD<A | B> r; const double ib = 1.0 / b.v; r.v = a.v * ib;
#define OP(k) if constexpr (SD_HAS(A | B, k)) {                                                                                          \
        if constexpr (SD_HAS(A, k) && SD_HAS(B, k)) r.d[sd_ix(A | B, k)] = (a.d[sd_ix(A, k)] - r.v * b.d[sd_ix(B, k)]) * ib;              \
        else if constexpr (SD_HAS(A, k)) r.d[sd_ix(A | B, k)] = a.d[sd_ix(A, k)] * ib;                                                    \
        else r.d[sd_ix(A | B, k)] = -(r.v * b.d[sd_ix(B, k)]) * ib; }
    SD_BITS(OP)
#undef OP
    return r;
}
template <unsigned M> __device__ __forceinline__ SD<M> operator-(const SD<M> &a) { SD<M> r; r.v = -a.v; _Pragma("unroll") for (int i = 0; i < sd_pc(M); i++) r.d[i] = -a.d[i]; return r; }
template <unsigned M> __device__ __forceinline__ SD<M> operator+(const SD<M> &a, double c) { SD<M> r = a; r.v += c; return r; }
template <unsigned M> __device__ __forceinline__ SD<M> operator+(double c, const SD<M> &a) { return a + c; }
template <unsigned M> __device__ __forceinline__ SD<M> operator-(const SD<M> &a, double c) { SD<M> r = a; r.v -= c; return r; }
template <unsigned M> __device__ __forceinline__ SD<M> operator-(double c, const SD<M> &a) { return (-a) + c; }
template <unsigned M> __device__ __forceinline__ SD<M> operator*(const SD<M> &a, double c) { SD<M> r; r.v = a.v * c; _Pragma("unroll") for (int i = 0; i < sd_pc(M); i++) r.d[i] = a.d[i] * c; return r; }
template <unsigned M> __device__ __forceinline__ SD<M> operator*(double c, const SD<M> &a) { return a * c; }
template <unsigned M> __device__ __forceinline__ SD<M> operator/(const SD<M> &a, double c) { return a * (1.0 / c); }
template <unsigned M> __device__ __forceinline__ SD<M> operator/(double c, const SD<M> &a)
{
    SD<M> r; const double ia = 1.0 / a.v; r.v = c * ia; const double f = -r.v * ia;
    _Pragma("unroll") for (int i = 0; i < sd_pc(M); i++) r.d[i] = f * a.d[i];
    return r;
}
#define SD_UNARY(name, fv, dfv)                                                                   \
    template <unsigned M> __device__ __forceinline__ SD<M> name(const SD<M> &a)                  \
    {                                                                                             \
        SD<M> r; const double x = a.v; const double f = (fv); const double df = (dfv); (void)f;    \
        r.v = f;                                                                                  \
        _Pragma("unroll") for (int i = 0; i < sd_pc(M); i++) r.d[i] = df * a.d[i];                \
        return r;                                                                                 \
    }
SD_UNARY(matan, nl_atan(x), 1.0 / fma(x, x, 1.0))
SD_UNARY(mtanh, nl_tanh(x), fma(-f, f, 1.0))
SD_UNARY(msqrt, sqrt(x), 0.5 / f)
#undef SD_UNARY
// sine with its derivative factor from the same range reduction
template <unsigned M> __device__ __forceinline__ SD<M> msin(const SD<M> &a)
{
    SD<M> r; const double2 sc = nl_sincos(a.v); r.v = sc.x;
    _Pragma("unroll") for (int i = 0; i < sd_pc(M); i++) r.d[i] = sc.y * a.d[i];
    return r;
}
template <unsigned M> __device__ __forceinline__ void msincos(const SD<M> &a, SD<M> &s, SD<M> &c)
{
    const double2 sc = nl_sincos(a.v); s.v = sc.x; c.v = sc.y;
    _Pragma("unroll") for (int i = 0; i < sd_pc(M); i++) { s.d[i] = sc.y * a.d[i]; c.d[i] = -sc.x * a.d[i]; }
}
__device__ __forceinline__ void msincos(double a, double &s, double &c) { const double2 sc = nl_sincos(a); s = sc.x; c = sc.y; }
// f(a, b) from its value and its two partial derivatives
template <unsigned A, unsigned B> __device__ __forceinline__ SD<A | B> sd_chain2(double f, double fa, double fb, const SD<A> &a, const SD<B> &b)
{
    SD<A | B> r; r.v = f;
#define OP(k) if constexpr (SD_HAS(A | B, k)) {                                                                                  \
        if constexpr (SD_HAS(A, k) && SD_HAS(B, k)) r.d[sd_ix(A | B, k)] = fma(fa, a.d[sd_ix(A, k)], fb * b.d[sd_ix(B, k)]);      \
        else if constexpr (SD_HAS(A, k)) r.d[sd_ix(A | B, k)] = fa * a.d[sd_ix(A, k)];                                            \
        else r.d[sd_ix(A | B, k)] = fb * b.d[sd_ix(B, k)]; }
    SD_BITS(OP)
#undef OP
    return r;
}
template <unsigned A, unsigned B> __device__ __forceinline__ SD<A | B> slip_angle_t(const SD<A> &v_lat, const SD<B> &v_lon)
{
    const SlipD sl = nl_slip_d(v_lat.v, v_lon.v);
    return sd_chain2(sl.a, sl.d_lat, sl.d_lon, v_lat, v_lon);
}
__device__ __forceinline__ double slip_angle_t(double v_lat, double v_lon) { return nl_slip(v_lat, v_lon); }
template <unsigned M> __device__ __forceinline__ SD<M> lat_pacejka_t(const SD<M> &alpha)
{
    const PacD pc = nl_pacejka_d(alpha.v);
    SD<M> r; r.v = pc.g;
    _Pragma("unroll") for (int i = 0; i < sd_pc(M); i++) r.d[i] = pc.dg * alpha.d[i];
    return r;
}
__device__ __forceinline__ double lat_pacejka_t(double alpha) { return nl_pacejka(alpha); }
template <unsigned M> __device__ __forceinline__ double val(const SD<M> &a) { return a.v; }
// compile-time choice between two expressions of different types
template <bool FIRST, typename TA, typename TB> __device__ __forceinline__ auto sd_sel(const TA &a, const TB &b)
{
    if constexpr (FIRST) return a; else return b;
}
__device__ __forceinline__ void sd_assign(double &dst, double src) { dst = src; }
template <unsigned TO, unsigned M> __device__ __forceinline__ void sd_assign(SD<TO> &dst, const SD<M> &src) { dst = sd_widen<TO>(src); }

// ---- fdyn6: Frenet 4-wheel Pacejka model (python/models.py:455-606), explicit form ----
// The tyre/chassis force model maps (v_x, v_y, r, T, delta) to (v_x_dot, v_y_dot, r_dot).  It is written once over scalar types:
// double for the plant, sparse duals seeded on the five inputs for the OCP (15 Jacobian entries by forward AD).
template <bool UNCROSSED, bool FENCE = false, typename TVX, typename TVY, typename TR, typename TT, typename TD, typename TO>
__device__ inline void fdyn6_forces(const TVX &v_x, const TVY &v_y, const TR &r, const TT &Tq, const TD &delta, TO &vxd, TO &vyd, TO &rd)
{
    // (scheduling fences between the wheels: left to itself the scheduler interleaves the four wheels' slip angles and Pacejka curves -- with their
    // partial derivatives some 150 values in flight -- and the linearisation kernel pays in scratch: 512 -> 400 B per lane, 6.4 -> 5.9 ms at 8192 x 40)
#define WHEEL_FENCE() do { if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0); } while (0)
    TD sd, cd;
    msincos(delta, sd, cd);
    const auto F_down = v_x * v_x * (0.5 * k_Cdown);
    const double cx = 0.5 * k_m * k_zCG / k_wheelbase, cy = 0.5 * k_m * k_zCG / k_axle_track;
    const auto base = F_down * 0.25 + k_static_weight;
    const double hx = 0.5 * k_axle_track;
    const auto v_x_FL = v_x - r * hx, v_x_FR = v_x + r * hx;
    const auto v_y_F = v_y + r * k_lF;
    const auto v_lon_FL = cd * v_x_FL + sd * v_y_F, v_lon_FR = cd * v_x_FR + sd * v_y_F;
    const auto v_lat_FL = cd * v_y_F - sd * v_x_FL, v_lat_FR = cd * v_y_F - sd * v_x_FR;
    const auto v_lat_R = v_y - r * k_lR;
    // slip angles: atan2(y, x) with x = smooth_abs_nonzero(.) > 0  ->  atan(y / x)
    const auto a_FL = slip_angle_t(v_lat_FL, v_lon_FL); WHEEL_FENCE();
    const auto a_FR = slip_angle_t(v_lat_FR, v_lon_FR); WHEEL_FENCE();
    const auto a_RL = slip_angle_t(v_lat_R, v_x_FL); WHEEL_FENCE();      // v_lon_RL = v_x - hx r
    const auto a_RR = slip_angle_t(v_lat_R, v_x_FR); WHEEL_FENCE();      // v_lon_RR = v_x + hx r
    // crossed slip angles exactly as models.py:543-546 (quirk Q3); order FL, FR, RL, RR.  UNCROSSED (model "fdyn6u") gives every
    // wheel its own slip angle: the crossed form is open-loop unstable (yaw eigenvalue +34 1/s at 10 m/s)
    const auto glat0 = lat_pacejka_t(sd_sel<UNCROSSED>(a_FL, a_RR)); WHEEL_FENCE();
    const auto glat1 = lat_pacejka_t(sd_sel<UNCROSSED>(a_FR, a_RL)); WHEEL_FENCE();
    const auto glat2 = lat_pacejka_t(sd_sel<UNCROSSED>(a_RL, a_FR)); WHEEL_FENCE();
    const auto glat3 = lat_pacejka_t(sd_sel<UNCROSSED>(a_RR, a_FL)); WHEEL_FENCE();
    const auto F_drag = -((v_x * v_x * k_Cr2 + v_x * k_Cr1 + k_Cr0) * mtanh(v_x * 10.0));
    const auto beta = matan((sd / cd) * k_rwd);                            // tan(delta) = sin / cos
    const auto r_kin = msqrt(v_x * v_x + v_y * v_y) * msin(beta) * (1.0 / k_lR);
    const auto dtau = (r_kin - r) * k_Ktv;
    const auto idenom = k_Cm0 / (F_down * (-0.25) - k_m * k_g);
    const auto gm = (Tq - dtau) * idenom, gp = (Tq + dtau) * idenom;      // glon: FL, RL = gm ; FR, RR = gp
   
    const auto cx0 = gm * cd - glat0 * sd, cy0 = gm * sd + glat0 * cd;   // FL
    const auto cx1 = gp * cd - glat1 * sd, cy1 = gp * sd + glat1 * cd;   // FR
    const auto cz0 = cy0 * k_lF - cx0 * hx, cz1 = cx1 * hx + cy1 * k_lF;
    const auto cz2 = -(gm * hx) - glat2 * k_lR, cz3 = gp * hx - glat3 * k_lR;
    // m a_x = X0 + Xx a_x + Xy a_y ; m a_y = Y0 + Yx a_x + Yy a_y with F_z,k = -(base + sx_k cx a_x + sy_k cy a_y),
    // sx = (-,-,+,+), sy = (+,-,+,-)
    const auto sumx = cx0 + cx1 + gm + gp, sumy = cy0 + cy1 + glat2 + glat3;
    const auto X0 = F_drag - sumx * base, Y0 = -(sumy * base);
    const auto Xx = (cx0 + cx1 - gm - gp) * cx, Xy = (cx1 - cx0 + gp - gm) * cy;
    const auto Yx = (cy0 + cy1 - glat2 - glat3) * cx, Yy = (cy1 - cy0 + glat3 - glat2) * cy;
   
    const auto a11 = -Xx + k_m, a12 = -Xy, a21 = -Yx, a22 = -Yy + k_m;
    const auto det = a11 * a22 - a12 * a21;
    const auto a_x = (X0 * a22 - a12 * Y0) / det;
    const auto a_y = (a11 * Y0 - a21 * X0) / det;
   
    const auto lx = a_x * cx, ly = a_y * cy;
    const auto Fz0 = -(base - lx + ly), Fz1 = -(base - lx - ly), Fz2 = -(base + lx + ly), Fz3 = -(base + lx - ly);
    const auto Mz = cz0 * Fz0 + cz1 * Fz1 + cz2 * Fz2 + cz3 * Fz3;
    sd_assign(vxd, a_x + v_y * r);
    sd_assign(vyd, a_y - v_x * r);
    sd_assign(rd, Mz * (1.0 / k_Iz));
}
#undef WHEEL_FENCE

// ---- the force model WITH its Jacobian on a ROW of sixteen lanes: the collocation plants (python/main.py:395-400) ----
// The collocation step has one lane per stage (a quad); the plant of ONE car leaves the other quads of a 16-lane row with nothing of their own to do,
// so the four quads of a row take one wheel each (w = (lane >> 2) & 3: FL, FR, RL, RR) for the slip angle and the Pacejka curve -- with their partial
// derivatives, five slots -- and fetch the other wheels' coefficients from the lanes of the same stage (ds_bpermute).  Rear-wheel quantities are widened
// to the front wheels' derivative mask (exact zeros in the steering slot): same values, same Jacobian as fdyn6_forces over the sparse duals.
template <unsigned M> __device__ __forceinline__ SD<M> sd_pick4(const int w, const SD<M> &a0, const SD<M> &a1, const SD<M> &a2, const SD<M> &a3)
{
    SD<M> r;
    r.v = (w == 0) ? a0.v : (w == 1) ? a1.v : (w == 2) ? a2.v : a3.v;
    _Pragma("unroll") for (int i = 0; i < sd_pc(M); i++) r.d[i] = (w == 0) ? a0.d[i] : (w == 1) ? a1.d[i] : (w == 2) ? a2.d[i] : a3.d[i];
    return r;
}
template <unsigned M> __device__ __forceinline__ SD<M> sd_row_get(const SD<M> &a, const int src_lane)
{
    SD<M> r;
    r.v = __shfl(a.v, src_lane);
    _Pragma("unroll") for (int i = 0; i < sd_pc(M); i++) r.d[i] = __shfl(a.d[i], src_lane);
    return r;
}
template <bool UNCROSSED, typename TVX, typename TVY, typename TR, typename TT, typename TD, typename TO>
__device__ inline void fdyn6_forces_row(const TVX &v_x, const TVY &v_y, const TR &r, const TT &Tq, const TD &delta, TO &vxd, TO &vyd, TO &rd)
{
    constexpr unsigned WM = 0x17u;      // v_x, v_y, r, delta: what a front wheel's slip angle depends on
    const int lane = threadIdx.x & 63, w = (lane >> 2) & 3, base_lane = lane & ~12;
    TD sd, cd;
    msincos(delta, sd, cd);
    const auto F_down = v_x * v_x * (0.5 * k_Cdown);
    const double cx = 0.5 * k_m * k_zCG / k_wheelbase, cy = 0.5 * k_m * k_zCG / k_axle_track;
    const auto base = F_down * 0.25 + k_static_weight;
    const double hx = 0.5 * k_axle_track;
    const auto v_x_FL = v_x - r * hx, v_x_FR = v_x + r * hx;
    const auto v_y_F = v_y + r * k_lF;
    const auto v_lon_FL = cd * v_x_FL + sd * v_y_F, v_lon_FR = cd * v_x_FR + sd * v_y_F;
    const auto v_lat_FL = cd * v_y_F - sd * v_x_FL, v_lat_FR = cd * v_y_F - sd * v_x_FR;
    const auto v_lat_R = v_y - r * k_lR;
    const SD<WM> v_lat_w = sd_pick4(w, sd_widen<WM>(v_lat_FL), sd_widen<WM>(v_lat_FR), sd_widen<WM>(v_lat_R), sd_widen<WM>(v_lat_R));
    const SD<WM> v_lon_w = sd_pick4(w, sd_widen<WM>(v_lon_FL), sd_widen<WM>(v_lon_FR), sd_widen<WM>(v_x_FL), sd_widen<WM>(v_x_FR));
    const SD<WM> a_w = slip_angle_t(v_lat_w, v_lon_w);
    // crossed slip angles (quirk Q3): wheel k takes the angle of wheel 3 - k
    const SD<WM> a_in = UNCROSSED ? a_w : sd_row_get(a_w, base_lane | ((3 - w) << 2));
    const SD<WM> g_w = lat_pacejka_t(a_in);
    const SD<WM> glat0 = sd_row_get(g_w, base_lane), glat1 = sd_row_get(g_w, base_lane | 4), glat2 = sd_row_get(g_w, base_lane | 8), glat3 = sd_row_get(g_w, base_lane | 12);
    const auto F_drag = -((v_x * v_x * k_Cr2 + v_x * k_Cr1 + k_Cr0) * mtanh(v_x * 10.0));
    const auto beta = matan((sd / cd) * k_rwd);
    const auto r_kin = msqrt(v_x * v_x + v_y * v_y) * msin(beta) * (1.0 / k_lR);
    const auto dtau = (r_kin - r) * k_Ktv;
    const auto idenom = k_Cm0 / (F_down * (-0.25) - k_m * k_g);
    const auto gm = (Tq - dtau) * idenom, gp = (Tq + dtau) * idenom;
    const auto cx0 = gm * cd - glat0 * sd, cy0 = gm * sd + glat0 * cd;
    const auto cx1 = gp * cd - glat1 * sd, cy1 = gp * sd + glat1 * cd;
    const auto cz0 = cy0 * k_lF - cx0 * hx, cz1 = cx1 * hx + cy1 * k_lF;
    const auto cz2 = -(gm * hx) - glat2 * k_lR, cz3 = gp * hx - glat3 * k_lR;
    const auto sumx = cx0 + cx1 + gm + gp, sumy = cy0 + cy1 + glat2 + glat3;
    const auto X0 = F_drag - sumx * base, Y0 = -(sumy * base);
    const auto Xx = (cx0 + cx1 - gm - gp) * cx, Xy = (cx1 - cx0 + gp - gm) * cy;
    const auto Yx = (cy0 + cy1 - glat2 - glat3) * cx, Yy = (cy1 - cy0 + glat3 - glat2) * cy;
    const auto a11 = -Xx + k_m, a12 = -Xy, a21 = -Yx, a22 = -Yy + k_m;
    const auto det = a11 * a22 - a12 * a21;
    const auto a_x = (X0 * a22 - a12 * Y0) / det;
    const auto a_y = (a11 * Y0 - a21 * X0) / det;
    const auto lx = a_x * cx, ly = a_y * cy;
    const auto Fz0 = -(base - lx + ly), Fz1 = -(base - lx - ly), Fz2 = -(base + lx + ly), Fz3 = -(base + lx - ly);
    const auto Mz = cz0 * Fz0 + cz1 * Fz1 + cz2 * Fz2 + cz3 * Fz3;
    sd_assign(vxd, a_x + v_y * r);
    sd_assign(vyd, a_y - v_x * r);
    sd_assign(rd, Mz * (1.0 / k_Iz));
}

// ---- the plant's force model on FOUR LANES: one wheel per lane ----
// A plant step is one car on one lane for 100 RK4 sub-steps (python/main.py:395-400: 100 steps per control period) -- a chain nothing else of the
// car can overlap, and four fifths of an evaluation are the four wheels' slip angles and Pacejka curves (two calls per wheel, ~570 instructions).
// The lanes q = 0..3 of a quad take one wheel each (FL, FR, RL, RR) and hand the coefficients round by DPP quad broadcasts; everything else every
// lane computes alike.  The same functions on the same operands as fdyn6_forces over doubles: BIT-IDENTICAL results, 2.5 x fewer instructions on
// the chain.  All four lanes of the quad must be active.
template <int CTRL>
__device__ __forceinline__ double quad_perm_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <bool UNCROSSED>
__device__ inline void fdyn6_forces_quad(const int q, const double v_x, const double v_y, const double r, const double Tq, const double delta,
                                         double &vxd, double &vyd, double &rd)
{
    double sd, cd;
    msincos(delta, sd, cd);
    const double F_down = v_x * v_x * (0.5 * k_Cdown);
    const double cx = 0.5 * k_m * k_zCG / k_wheelbase, cy = 0.5 * k_m * k_zCG / k_axle_track;
    const double base = F_down * 0.25 + k_static_weight;
    const double hx = 0.5 * k_axle_track;
    const double v_x_FL = v_x - r * hx, v_x_FR = v_x + r * hx;
    const double v_y_F = v_y + r * k_lF;
    const double v_lon_FL = cd * v_x_FL + sd * v_y_F, v_lon_FR = cd * v_x_FR + sd * v_y_F;
    const double v_lat_FL = cd * v_y_F - sd * v_x_FL, v_lat_FR = cd * v_y_F - sd * v_x_FR;
    const double v_lat_R = v_y - r * k_lR;
    // this lane's wheel: slip angle, then the Pacejka coefficient on its own angle (fdyn6u) or on the opposite wheel's (as written, quirk Q3)
    const double v_lat_q = (q == 0) ? v_lat_FL : (q == 1) ? v_lat_FR : v_lat_R;
    const double v_lon_q = (q == 0) ? v_lon_FL : (q == 1) ? v_lon_FR : (q == 2) ? v_x_FL : v_x_FR;
    const double a_q = slip_angle_t(v_lat_q, v_lon_q);
    const double g_q = lat_pacejka_t(UNCROSSED ? a_q : quad_perm_f64<0x1B>(a_q));          // quad_perm [3,2,1,0]: wheel k reads wheel 3 - k
    const double glat0 = quad_perm_f64<0x00>(g_q), glat1 = quad_perm_f64<0x55>(g_q), glat2 = quad_perm_f64<0xAA>(g_q), glat3 = quad_perm_f64<0xFF>(g_q);
    const double F_drag = -((v_x * v_x * k_Cr2 + v_x * k_Cr1 + k_Cr0) * mtanh(v_x * 10.0));
    const double beta = matan((sd / cd) * k_rwd);
    const double r_kin = msqrt(v_x * v_x + v_y * v_y) * msin(beta) * (1.0 / k_lR);
    const double dtau = (r_kin - r) * k_Ktv;
    const double idenom = k_Cm0 / (F_down * (-0.25) - k_m * k_g);
    const double gm = (Tq - dtau) * idenom, gp = (Tq + dtau) * idenom;
    const double cx0 = gm * cd - glat0 * sd, cy0 = gm * sd + glat0 * cd;
    const double cx1 = gp * cd - glat1 * sd, cy1 = gp * sd + glat1 * cd;
    const double cz0 = cy0 * k_lF - cx0 * hx, cz1 = cx1 * hx + cy1 * k_lF;
    const double cz2 = -(gm * hx) - glat2 * k_lR, cz3 = gp * hx - glat3 * k_lR;
    const double sumx = cx0 + cx1 + gm + gp, sumy = cy0 + cy1 + glat2 + glat3;
    const double X0 = F_drag - sumx * base, Y0 = -(sumy * base);
    const double Xx = (cx0 + cx1 - gm - gp) * cx, Xy = (cx1 - cx0 + gp - gm) * cy;
    const double Yx = (cy0 + cy1 - glat2 - glat3) * cx, Yy = (cy1 - cy0 + glat3 - glat2) * cy;
    const double a11 = -Xx + k_m, a12 = -Xy, a21 = -Yx, a22 = -Yy + k_m;
    const double det = a11 * a22 - a12 * a21;
    const double a_x = (X0 * a22 - a12 * Y0) / det;
    const double a_y = (a11 * Y0 - a21 * X0) / det;
    const double lx = a_x * cx, ly = a_y * cy;
    const double Fz0 = -(base - lx + ly), Fz1 = -(base - lx - ly), Fz2 = -(base + lx + ly), Fz3 = -(base + lx - ly);
    const double Mz = cz0 * Fz0 + cz1 * Fz1 + cz2 * Fz2 + cz3 * Fz3;
    vxd = a_x + v_y * r;
    vyd = a_y - v_x * r;
    rd = Mz * (1.0 / k_Iz);
}
// xdot of the dynamic plant on the four lanes of a quad (no Jacobian): fdyn6_eval<false, UNCROSSED> with the wheels spread over the lanes
template <bool UNCROSSED>
__device__ inline void fdyn6_eval_quad(const int q, const double (&x)[8], double u_T, double u_delta, TrackSeg &trk, double (&f)[8])
{
    const double n = x[1], psi = x[2], v_x = x[3], v_y = x[4], r = x[5], T = x[6], delta = x[7];
    double dk;
    const double kap = trk.kappa(x[0], dk);
    double sp, cp;
    fast_sincos(psi, &sp, &cp);
    const double num = v_x * cp - v_y * sp;
    const double inv_dn = 1.0 / (1.0 + kap * n);
    const double s_dot = num * inv_dn;
    f[0] = s_dot;
    f[1] = v_x * sp + v_y * cp;
    f[2] = r - kap * s_dot;
    f[6] = (u_T - T) * (1.0 / k_tT);
    f[7] = (u_delta - delta) * (1.0 / k_tdelta);
    fdyn6_forces_quad<UNCROSSED>(q, v_x, v_y, r, T, delta, f[3], f[4], f[5]);
}

// xdot (and with WITH_JAC the structural non-zeros of its Jacobian, pattern JX_MASK[1] / JU_MASK[1])
// ROW (collocation plants only, all sixteen lanes of a row on the same car and active): the wheels spread over the row's quads (fdyn6_forces_row)
template <bool WITH_JAC, bool UNCROSSED, bool ROW = false, bool FENCE = false>
__device__ inline void fdyn6_eval(const double (&x)[8], double u_T, double u_delta, TrackSeg &trk, double (&f)[8], double (&J)[8][10])
{
    const double n = x[1], psi = x[2], v_x = x[3], v_y = x[4], r = x[5], T = x[6], delta = x[7];
    double dk;
    const double kap = trk.kappa(x[0], dk);
    double sp, cp;
    fast_sincos(psi, &sp, &cp);
    const double num = v_x * cp - v_y * sp;
    const double inv_dn = 1.0 / (1.0 + kap * n);
    const double s_dot = num * inv_dn;
    f[0] = s_dot;
    f[1] = v_x * sp + v_y * cp;
    f[2] = r - kap * s_dot;
    f[6] = (u_T - T) * (1.0 / k_tT);
    f[7] = (u_delta - delta) * (1.0 / k_tdelta);
    if (WITH_JAC) {
        SD<0x1Fu> o0, o1, o2;
        if constexpr (ROW) fdyn6_forces_row<UNCROSSED>(sd_seed<1u>(v_x), sd_seed<2u>(v_y), sd_seed<4u>(r), sd_seed<8u>(T), sd_seed<16u>(delta), o0, o1, o2);
        else fdyn6_forces<UNCROSSED, FENCE>(sd_seed<1u>(v_x), sd_seed<2u>(v_y), sd_seed<4u>(r), sd_seed<8u>(T), sd_seed<16u>(delta), o0, o1, o2);
        f[3] = o0.v; f[4] = o1.v; f[5] = o2.v;
#pragma unroll
        for (int c = 0; c < 5; c++) { J[3][3 + c] = o0.d[c]; J[4][3 + c] = o1.d[c]; J[5][3 + c] = o2.d[c]; }
        const double q = -s_dot * inv_dn;
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
        J[6][6] = -1.0 / k_tT;
        J[6][8] = 1.0 / k_tT;
        J[7][7] = -1.0 / k_tdelta;
        J[7][9] = 1.0 / k_tdelta;
    } else {
        fdyn6_forces<UNCROSSED>(v_x, v_y, r, T, delta, f[3], f[4], f[5]);
    }
}

#pragma clang fp contract(fast)

}  // namespace ihm2
