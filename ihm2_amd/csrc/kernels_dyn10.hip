// kernels_dyn10.hip -- the DYN10 plant of the reference's MiL loop with the reference's integrator, usable from rest.
//
//   model       fdyn10_model, python/models.py:609-801 (15 states, 5 inputs; implicit residual there, solved for xdot here through the
//               2 x 2 load-transfer system, as k_sim_dyn10 in kernels_cart.hip)
//   integrator  AcadosSimOpts of python/main.py:395-400: IRK, GAUSS_RADAU_IIA, 4 stages, 100 steps over dt
//   call site   python/main.py:438-441 (x0 = (-6, 0, ..., 0): the car starts at REST), :490-502 (the DYN10 plant step)
//
// At standstill the slip ratios omega R_w / smooth_abs_nonzero(v) - 1 make the wheel dynamics stiff (1e9 1/s) with a basin of
// attraction of 1e-7 rad/s around the rolling condition: a fixed number of Newton iterations from K = 0 (acados' default: 3) does not
// converge there.  The collocation equations are therefore solved TO CONVERGENCE and the step length follows the Newton iteration
// (predictor = the derivative at the end of the previous step; a step that does not converge is retried with h / 4; a step that
// converges in at most four iterations lets the next one try 2 h, up to the reference's grid dt / M).  The same algorithm on the CPU
// (the tests' reference restatement) agrees with scipy's adaptive Radau at rtol 1e-12 to 1e-13 from rest.
//
// Mapping: ONE WAVEFRONT PER CAR.  The Newton matrix I - h (A (x) J_i) of the four stages is 60 x 60 and dense in the model's
// coupling (velocities, wheel speeds and load transfer): it lives in LDS (60 x 61 doubles incl. the right-hand side, rows padded to an
// odd stride), lane (i, a) = 15 i + a evaluates the model at stage i over dual numbers seeded in state direction a -- value = f_i,
// derivative = column a of J_i -- and the elimination with partial pivoting runs one COLUMN per lane (conflict-free LDS rows, the
// pivot row and the multipliers are broadcasts).  Pivot rule: the first row with the largest magnitude.
#include <hip/hip_runtime.h>

#include "ihm2mpc_internal.h"
#include "model.hpp"

using namespace ihm2;

namespace {

// value + one directional derivative
struct D1 {
    double v, d;
};
__device__ __forceinline__ D1 mk(double v, double d = 0.0) { D1 r; r.v = v; r.d = d; return r; }
__device__ __forceinline__ D1 operator+(D1 a, D1 b) { return mk(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ D1 operator-(D1 a, D1 b) { return mk(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ D1 operator-(D1 a) { return mk(-a.v, -a.d); }
__device__ __forceinline__ D1 operator*(D1 a, D1 b) { return mk(a.v * b.v, a.v * b.d + a.d * b.v); }
__device__ __forceinline__ D1 operator/(D1 a, D1 b) { const double q = a.v / b.v; return mk(q, (a.d - q * b.d) / b.v); }
__device__ __forceinline__ D1 operator+(D1 a, double c) { return mk(a.v + c, a.d); }
__device__ __forceinline__ D1 operator+(double c, D1 a) { return mk(a.v + c, a.d); }
__device__ __forceinline__ D1 operator-(D1 a, double c) { return mk(a.v - c, a.d); }
__device__ __forceinline__ D1 operator-(double c, D1 a) { return mk(c - a.v, -a.d); }
__device__ __forceinline__ D1 operator*(D1 a, double c) { return mk(a.v * c, a.d * c); }
__device__ __forceinline__ D1 operator*(double c, D1 a) { return mk(a.v * c, a.d * c); }
__device__ __forceinline__ D1 operator/(D1 a, double c) { return mk(a.v / c, a.d / c); }
__device__ __forceinline__ D1 d_sin(D1 a) { double s, c; sincos(a.v, &s, &c); return mk(s, a.d * c); }
__device__ __forceinline__ D1 d_cos(D1 a) { double s, c; sincos(a.v, &s, &c); return mk(c, -a.d * s); }
__device__ __forceinline__ D1 d_atan(D1 a) { return mk(atan(a.v), a.d / (1.0 + a.v * a.v)); }
__device__ __forceinline__ D1 d_tanh(D1 a) { const double t = tanh(a.v); return mk(t, a.d * (1.0 - t * t)); }
__device__ __forceinline__ D1 d_exp(D1 a) { const double e = exp(a.v); return mk(e, a.d * e); }
// atan2(y, x) with x > 0 (x = smooth_abs_nonzero(.) > 0 at the call sites)
__device__ __forceinline__ D1 d_atan2_pos(D1 y, D1 x) { return mk(atan2(y.v, x.v), (x.v * y.d - y.v * x.d) / (x.v * x.v + y.v * y.v)); }
__device__ __forceinline__ D1 d_sabs_nz(D1 v) { return d_tanh(10.0 * v) * v + 1e-6 * d_exp(-(v * v)); }      // python/utils.py:27-28

constexpr double q_b1s = -6.75e-6, q_b2s = 1.35e-1, q_b3s = 1.2e-3, q_c1s = 1.86, q_d1s = 1.12e-4, q_d2s = 1.57, q_e1s = -5.38e-6, q_e2s = 1.11e-2, q_e3s = -4.26;
constexpr double q_Rw = 0.20809, q_Iw = 0.3, q_kd = 0.17, q_ks = 15.0;

// xdot = f(x, u) of fdyn10 over dual numbers (python/models.py:609-801 solved for xdot; wheel order FL, FR, RL, RR)
__device__ void fdyn10_dual(const D1 (&x)[15], const double (&u)[5], TrackSeg &trk, D1 (&f)[15])
{
    const D1 n = x[1], psi = x[2], v_x = x[3], v_y = x[4], r = x[5], delta = x[14];
    const double W0 = k_static_weight;
    const double BCDs = (q_b1s * W0 * W0 + q_b2s * W0) * exp(-q_b3s * W0), Cs = q_c1s, Ds = q_d1s * W0 + q_d2s, Es = q_e1s * W0 * W0 + q_e2s * W0 + q_e3s;
    const double Bs = BCDs / (Cs * Ds);
    const double BCDa = k_b1a * sin(2.0 * atan(W0 / k_b2a)), Ca = k_c1a, Da = k_d1a * W0 + k_d2a, Ea = k_e1a * W0 + k_e2a, Ba = BCDa / (Ca * Da);
    const D1 sd = d_sin(delta), cd = d_cos(delta);
    const D1 F_drag = -((k_Cr0 + k_Cr1 * v_x + k_Cr2 * (v_x * v_x)) * d_tanh(1000.0 * v_x));
    const D1 base = W0 + 0.25 * (0.5 * k_Cdown * (v_x * v_x));
    const double cx = 0.5 * k_m * k_zCG / k_wheelbase, cy = 0.5 * k_m * k_zCG / k_axle_track, hx = 0.5 * k_axle_track;
    const D1 vxL = v_x - hx * r, vxR = v_x + hx * r, vyF = v_y + k_lF * r, vyR = v_y - k_lR * r;
    const D1 v_lon[4] = {cd * vxL + sd * vyF, cd * vxR + sd * vyF, vxL, vxR};
    const D1 v_lat[4] = {cd * vyF - sd * vxL, cd * vyF - sd * vxR, vyR, vyR};
    D1 cs[4], fx[4], fy[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const D1 va = d_sabs_nz(v_lon[w]);
        const D1 Ba_a = Ba * d_atan2_pos(v_lat[w], va), Bs_s = Bs * (x[6 + w] * q_Rw / va - 1.0);
        const D1 cl = Da * d_sin(Ca * d_atan(Ba_a - Ea * (Ba_a - d_atan(Ba_a))));
        cs[w] = Ds * d_sin(Cs * d_atan(Bs_s - Es * (Bs_s - d_atan(Bs_s))));
        // body-frame force per unit of normal load N_w = -F_z,w:  F_lon = N cs, F_lat = -N cl
        if (w < 2) { fx[w] = cd * cs[w] + sd * cl; fy[w] = sd * cs[w] - cd * cl; }
        else { fx[w] = cs[w]; fy[w] = -cl; }
    }
    // N_w = base + sx_w cx a_x + sy_w cy a_y, sx = (-,-,+,+), sy = (+,-,+,-)
    const D1 Sfx = fx[0] + fx[1] + fx[2] + fx[3], Sfy = fy[0] + fy[1] + fy[2] + fy[3];
    const D1 Sxx = fx[2] + fx[3] - fx[0] - fx[1], Sxy = fx[0] - fx[1] + fx[2] - fx[3];
    const D1 Syx = fy[2] + fy[3] - fy[0] - fy[1], Syy = fy[0] - fy[1] + fy[2] - fy[3];
    const D1 a11 = k_m - cx * Sxx, a12 = -(cy * Sxy), a21 = -(cx * Syx), a22 = k_m - cy * Syy;
    const D1 b1 = F_drag + base * Sfx, b2 = base * Sfy, det = a11 * a22 - a12 * a21;
    const D1 a_x = (b1 * a22 - a12 * b2) / det, a_y = (a11 * b2 - a21 * b1) / det;
    const D1 lx = cx * a_x, ly = cy * a_y;
    const D1 Nw[4] = {base - lx + ly, base - lx - ly, base + lx + ly, base + lx - ly};
    double dk;
    const double kr = trk.kappa(x[0].v, dk);
    const D1 kap = mk(kr, x[0].d * dk);
    const D1 sp = d_sin(psi), cp = d_cos(psi);
    const D1 s_dot = (v_x * cp - v_y * sp) / (1.0 + kap * n);
    f[0] = s_dot;
    f[1] = v_x * sp + v_y * cp;
    f[2] = r - kap * s_dot;
    f[3] = a_x + v_y * r;
    f[4] = a_y - v_x * r;
    f[5] = ((Nw[1] * fx[1] - Nw[0] * fx[0]) * hx + (Nw[1] * fy[1] + Nw[0] * fy[0]) * k_lF + (Nw[3] * fx[3] - Nw[2] * fx[2]) * hx
            - (Nw[3] * fy[3] + Nw[2] * fy[2]) * k_lR) / k_Iz;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        f[6 + w] = (x[10 + w] - (q_kd * x[6 + w] + q_ks + q_Rw * (Nw[w] * cs[w]))) / q_Iw;
        f[10 + w] = (u[w] - x[10 + w]) / k_tT;
    }
    f[14] = (u[4] - delta) / k_tdelta;
}

struct Dyn10Tab {
    double A[4][4], b[4];
    int radau;      // the last stage sits at the end of the step (c_4 = 1): its derivative is the next step's predictor
};

constexpr int NXD = 15, NK = 60, LDM = 61 + 0;      // row stride of the augmented matrix [M | rhs] in doubles (61 columns; odd: lanes on a column hit 64 different banks)

// wave-wide (max |value|, smallest index among equals): the first row with the largest magnitude
__device__ __forceinline__ void wave_argmax(double &v, int &idx)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}
__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

__global__ __launch_bounds__(64) void k_sim_dyn10_irk(int B, int M, int newton_iter, double dt, Dyn10Tab tab, int nknots, const double *__restrict__ s_ref,
                                                      const double *__restrict__ kappa_ref, const int32_t *__restrict__ track_id, const double *xs,
                                                      const double *__restrict__ us, double *xn)
{
    __shared__ double Mx[NK * LDM];       // [M | rhs]: row r, column c at r * LDM + c; column 60 = right-hand side / Newton update
    __shared__ double K[4 * NXD];         // stage derivatives
    __shared__ double F[4 * NXD];         // f at the stage points
    __shared__ double xc[NXD], kend[NXD];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    const int st = min(lane / NXD, 3), dir = lane - st * NXD;        // lanes 60..63 ride along with stage 3 (direction >= 15: no seed, results unused)
    const bool owner = lane < NK;
    double u[5];
#pragma unroll
    for (int i = 0; i < 5; i++) u[i] = us[(size_t)b * 5 + i];
    if (lane < NXD) xc[lane] = xs[(size_t)b * NXD + lane];
    __syncthreads();
    const int tid = track_id[b];
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, xc[0]);

    // model at stage point X (value in xv) seeded along `dir`: f -> fv, column `dir` of the Jacobian -> jc
    auto eval = [&](const double (&xv)[NXD], double (&fv)[NXD], double (&jc)[NXD]) {
        D1 X[NXD], Fd[NXD];
#pragma unroll
        for (int a = 0; a < NXD; a++) X[a] = mk(xv[a], (a == dir) ? 1.0 : 0.0);
        fdyn10_dual(X, u, trk, Fd);
#pragma unroll
        for (int a = 0; a < NXD; a++) { fv[a] = Fd[a].v; jc[a] = Fd[a].d; }
    };

    {   // predictor of the first step: f(x0)
        double xv[NXD], fv[NXD], jc[NXD];
#pragma unroll
        for (int a = 0; a < NXD; a++) xv[a] = xc[a];
        eval(xv, fv, jc);
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < NXD; a++) kend[a] = fv[a];
        }
    }
    __syncthreads();

    const double h_max = dt / M, h_min = dt * 1e-12;
    double t = 0.0, h = h_max;
    bool failed = false;
    // every quantity that steers the loops below is wave-uniform (reductions end in every lane holding the same value)
    while (t < dt * (1.0 - 1e-14) && !failed) {
        if (h > dt - t) h = dt - t;
        if (owner) K[lane] = kend[dir];
        __syncthreads();
        bool conv = false;
        int it = 0;
        for (; it < newton_iter && !conv; it++) {
            // ---- stage points, model + Jacobian columns, assembly of [I - h (A (x) J_i) | -(K - F)] ----
            double xv[NXD], fv[NXD], jc[NXD];
#pragma unroll
            for (int a = 0; a < NXD; a++) {
                double acc = xc[a];
#pragma unroll
                for (int j = 0; j < 4; j++) acc += h * tab.A[st][j] * K[j * NXD + a];
                xv[a] = acc;
            }
            eval(xv, fv, jc);
            if (owner) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const double ha = h * tab.A[st][j];
#pragma unroll
                    for (int a = 0; a < NXD; a++)
                        Mx[(st * NXD + a) * LDM + j * NXD + dir] = ((st == j && a == dir) ? 1.0 : 0.0) - ha * jc[a];
                }
                if (dir == 0) {
#pragma unroll
                    for (int a = 0; a < NXD; a++) { F[st * NXD + a] = fv[a]; Mx[(st * NXD + a) * LDM + NK] = -(K[st * NXD + a] - fv[a]); }
                }
            }
            __syncthreads();
            // ---- elimination with partial pivoting on the augmented matrix: lane = column (lane 60 = right-hand side) ----
            bool singular = false;
            for (int c = 0; c < NK; c++) {
                double pv = (lane >= c && lane < NK) ? fabs(Mx[lane * LDM + c]) : -1.0;
                int pi = lane;
                wave_argmax(pv, pi);
                if (!(pv > 0.0)) { singular = true; break; }
                if (pi != c && lane <= NK) {
                    const double t0 = Mx[c * LDM + lane], t1 = Mx[pi * LDM + lane];
                    Mx[c * LDM + lane] = t1; Mx[pi * LDM + lane] = t0;
                }
                __syncthreads();
                const double piv = Mx[c * LDM + c];
                if (lane > c && lane <= NK) {
                    const double top = Mx[c * LDM + lane];
                    for (int r = c + 1; r < NK; r++) {
                        const double l = Mx[r * LDM + c] / piv;
                        Mx[r * LDM + lane] -= l * top;
                    }
                }
                __syncthreads();
            }
            if (singular) break;
            // ---- back substitution on the right-hand side column: lane = row ----
            for (int c = NK - 1; c >= 0; c--) {
                if (lane == c) Mx[c * LDM + NK] /= Mx[c * LDM + c];
                __syncthreads();
                if (lane < c) Mx[lane * LDM + NK] -= Mx[lane * LDM + c] * Mx[c * LDM + NK];
                __syncthreads();
            }
            // ---- Newton update and convergence test ----
            double d = 0.0;
            if (owner) {
                const double dk = Mx[lane * LDM + NK], kn = K[lane] + dk;
                K[lane] = kn;
                d = fabs(dk) / (1.0 + fabs(kn));
                if (!(d == d)) d = INFINITY;
            }
            d = wave_max_d(d);
            __syncthreads();
            if (!(d < INFINITY)) break;
            conv = d <= 1e-10;
        }
        if (!conv) {
            h *= 0.25;
            if (h < h_min) failed = true;
            continue;
        }
        __syncthreads();
        if (lane < NXD) {
            double acc = xc[lane];
#pragma unroll
            for (int i = 0; i < 4; i++) acc += h * tab.b[i] * K[i * NXD + lane];
            xc[lane] = acc;
        }
        __syncthreads();
        if (tab.radau) {
            if (lane < NXD) kend[lane] = K[3 * NXD + lane];
        } else {
            double xv[NXD], fv[NXD], jc[NXD];
#pragma unroll
            for (int a = 0; a < NXD; a++) xv[a] = xc[a];
            eval(xv, fv, jc);
            if (lane == 0) {
#pragma unroll
                for (int a = 0; a < NXD; a++) kend[a] = fv[a];
            }
        }
        __syncthreads();
        t += h;
        if (it <= 4) h = fmin(2.0 * h, h_max);
    }
    if (lane < NXD) xn[(size_t)b * NXD + lane] = failed ? NAN : xc[lane];
}

}  // namespace

#include "irk_tableaux.h"

// integ: IHM2MPC_INTEG_IRK_RADAU4 (the reference's plants) or IHM2MPC_INTEG_IRK_GL4
void ihm2_launch_sim_dyn10_irk(ihm2mpc_handle *h, int integ, int M, int newton_iter, const double *x, const double *u, double *xn, hipStream_t stream)
{
    Dyn10Tab tab;
    const bool radau = integ != IHM2MPC_INTEG_IRK_GL4;
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) tab.A[i][j] = radau ? IRK_RADAU4_A[i][j] : IRK_GL4_A[i][j];
        tab.b[i] = radau ? IRK_RADAU4_b[i] : IRK_GL4_b[i];
    }
    tab.radau = radau ? 1 : 0;
    hipLaunchKernelGGL(k_sim_dyn10_irk, dim3(h->B), dim3(64), 0, stream, h->B, M, newton_iter, h->cfg.dt, tab, h->cfg.nknots, h->s_ref, h->kappa_ref,
                       h->track_id, x, u, xn);
}
