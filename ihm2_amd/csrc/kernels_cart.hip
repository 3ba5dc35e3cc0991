// kernels_cart.hip -- the Cartesian side of the ROS stack, batched:
//  * k_sim_cart : plant steps of the simulation node (src/ihm2/src/sim_node.cpp:197-257): kin6 below v_dyn, dyn6 above
//                 (python/models.py:168-229, 310-452), RK4 x M per plant step, "no reversing" clamp;
//  * k_project  : Track::project (src/ihm2/src/common/tracks.cpp:183-288) and the Frenet states of the control node
//                 (src/ihm2/src/mpc_control_node.cpp:142-157).
// One lane per instance: both are a few hundred dependent operations on 8 doubles, off the solver's critical path
// (ihm2mpc_step runs the plant beside the linearisation).
#include "ihm2mpc_internal.h"
#include "model.hpp"

using namespace ihm2;

namespace {

__device__ __forceinline__ double lat_pacejka_d(double alpha) { return lat_pacejka_t(alpha); }

// kin6: the fkin6 force model with Cartesian kinematics and r_dot = v_y_dot / l_R (python/models.py:226)
__device__ inline void kin6_rhs(const double (&x)[8], double u_T, double u_delta, double (&f)[8])
{
    const double phi = x[2], v_x = x[3], v_y = x[4], r = x[5], T = x[6], delta = x[7];
    const double delta_dot = (u_delta - delta) * (1.0 / k_tdelta), T_dot = (u_T - T) * (1.0 / k_tT);
    const double F_motor = k_Cm0 * T;
    const double F_drag = -(k_Cr0 + k_Cr1 * v_x + k_Cr2 * v_x * v_x) * tanh(10.0 * v_x);
    const double F_Rx = 0.5 * F_motor + F_drag, F_Fx = 0.5 * F_motor;
    const double td = tan(delta);
    const double q = 1.0 + k_rwd * k_rwd * td * td, iden = rsqrt(q);
    const double sinbeta = k_rwd * td * iden, cosbeta = iden;
    const double beta = atan(k_rwd * td);
    const double beta_dot = k_rwd * (1.0 + td * td) / q * delta_dot;
    const double v_dot = (F_Rx * cosbeta + F_Fx * cos(delta - beta)) * (1.0 / k_m);
    double sp, cp;
    sincos(phi, &sp, &cp);
    const double v_y_dot = v_dot * sinbeta + beta_dot * v_x;
    f[0] = v_x * cp - v_y * sp;
    f[1] = v_x * sp + v_y * cp;
    f[2] = r;
    f[3] = v_dot * cosbeta - beta_dot * v_y;
    f[4] = v_y_dot;
    f[5] = v_y_dot * (1.0 / k_lR);
    f[6] = T_dot;
    f[7] = delta_dot;
}

// dyn6: 4-wheel Pacejka model in the body frame; the load transfer makes the tyre forces affine in (a_x, a_y): 2x2 solve
__device__ inline void dyn6_rhs(const double (&x)[8], double u_T, double u_delta, double (&f)[8])
{
    const double phi = x[2], v_x = x[3], v_y = x[4], r = x[5], T = x[6], delta = x[7];
    double sd, cd;
    sincos(delta, &sd, &cd);
    const double F_down = 0.5 * k_Cdown * v_x * v_x;
    const double cx = 0.5 * k_m * k_zCG / k_wheelbase, cy = 0.5 * k_m * k_zCG / k_axle_track;
    const double base = k_static_weight + 0.25 * F_down, hx = 0.5 * k_axle_track;
    const double v_x_L = v_x - hx * r, v_x_R = v_x + hx * r, v_y_F = v_y + k_lF * r, v_y_R = v_y - k_lR * r;
    const double glat0 = lat_pacejka_d(atan2(v_y_F, v_x_L) - delta), glat1 = lat_pacejka_d(atan2(v_y_F, v_x_R) - delta);
    const double glat2 = lat_pacejka_d(atan2(v_y_R, v_x_L)), glat3 = lat_pacejka_d(atan2(v_y_R, v_x_R));
    const double F_drag = -(k_Cr0 + k_Cr1 * v_x + k_Cr2 * v_x * v_x) * tanh(1000.0 * v_x);
    const double td = tan(delta);
    const double sinbeta = k_rwd * td * rsqrt(1.0 + k_rwd * k_rwd * td * td);
    const double dtau = k_Ktv * (v_x * sinbeta * (1.0 / k_lR) - r);
    const double denom = -k_m * k_g - 0.25 * F_down;
    const double gm = k_Cm0 * (T - dtau) / denom, gp = k_Cm0 * (T + dtau) / denom;      // FL, RL = gm ; FR, RR = gp
    const double cx0 = gm * cd - glat0 * sd, cy0 = gm * sd + glat0 * cd;
    const double cx1 = gp * cd - glat1 * sd, cy1 = gp * sd + glat1 * cd;
    const double cz0 = cy0 * k_lF - cx0 * hx, cz1 = cx1 * hx + cy1 * k_lF;
    const double cz2 = -(gm * hx) - glat2 * k_lR, cz3 = gp * hx - glat3 * k_lR;
    // m a_x = X0 + Xx a_x + Xy a_y ; m a_y = Y0 + Yx a_x + Yy a_y with F_z,k = -(base + sx_k cx a_x + sy_k cy a_y)
    const double sumx = cx0 + cx1 + gm + gp, sumy = cy0 + cy1 + glat2 + glat3;
    const double X0 = F_drag - sumx * base, Y0 = -(sumy * base);
    const double Xx = (cx0 + cx1 - gm - gp) * cx, Xy = (cx1 - cx0 + gp - gm) * cy;
    const double Yx = (cy0 + cy1 - glat2 - glat3) * cx, Yy = (cy1 - cy0 + glat3 - glat2) * cy;
    const double a11 = k_m - Xx, a12 = -Xy, a21 = -Yx, a22 = k_m - Yy;
    const double det = a11 * a22 - a12 * a21;
    const double a_x = (X0 * a22 - a12 * Y0) / det, a_y = (a11 * Y0 - a21 * X0) / det;
    const double lx = a_x * cx, ly = a_y * cy;
    const double Mz = cz0 * -(base - lx + ly) + cz1 * -(base - lx - ly) + cz2 * -(base + lx + ly) + cz3 * -(base + lx - ly);
    double sp, cp;
    sincos(phi, &sp, &cp);
    f[0] = v_x * cp - v_y * sp;
    f[1] = v_x * sp + v_y * cp;
    f[2] = r;
    f[3] = a_x + v_y * r;
    f[4] = a_y - v_x * r;
    f[5] = Mz * (1.0 / k_Iz);
    f[6] = (u_T - T) * (1.0 / k_tT);
    f[7] = (u_delta - delta) * (1.0 / k_tdelta);
}

// fdyn10 (python/models.py:609-801): 15 states (s, n, psi, v_x, v_y, r, four wheel speeds, four wheel torques, delta), 5 inputs.  The
// reference writes it as an implicit residual; the normal loads are affine in (a_x, a_y) and the tyre forces are normal load x Pacejka
// coefficient, so m a = F is a 2 x 2 linear system (as for dyn6 above).  Wheel order FL, FR, RL, RR.
constexpr double k_b1s = -6.75e-6, k_b2s = 1.35e-1, k_b3s = 1.2e-3, k_c1s = 1.86, k_d1s = 1.12e-4, k_d2s = 1.57, k_e1s = -5.38e-6, k_e2s = 1.11e-2, k_e3s = -4.26;
constexpr double k_Rw = 0.20809, k_Iw = 0.3, k_kd = 0.17, k_ks = 15.0;
__device__ inline void fdyn10_rhs(const double (&x)[15], const double (&u)[5], TrackSeg &trk, double (&f)[15])
{
    const double n = x[1], psi = x[2], v_x = x[3], v_y = x[4], r = x[5], delta = x[14];
    const double W0 = k_static_weight;
    const double BCDs = (k_b1s * W0 * W0 + k_b2s * W0) * exp(-k_b3s * W0), Cs = k_c1s, Ds = k_d1s * W0 + k_d2s, Es = k_e1s * W0 * W0 + k_e2s * W0 + k_e3s;
    const double Bs = BCDs / (Cs * Ds);
    double sd, cd;
    fast_sincos(delta, &sd, &cd);
    const double F_drag = -(k_Cr0 + k_Cr1 * v_x + k_Cr2 * v_x * v_x) * tanh_e(1000.0 * v_x);
    const double base = W0 + 0.25 * (0.5 * k_Cdown * v_x * v_x);
    const double cx = 0.5 * k_m * k_zCG / k_wheelbase, cy = 0.5 * k_m * k_zCG / k_axle_track, hx = 0.5 * k_axle_track;
    const double vxL = v_x - hx * r, vxR = v_x + hx * r, vyF = v_y + k_lF * r, vyR = v_y - k_lR * r;
    const double v_lon[4] = {cd * vxL + sd * vyF, cd * vxR + sd * vyF, vxL, vxR};
    const double v_lat[4] = {-sd * vxL + cd * vyF, -sd * vxR + cd * vyF, vyR, vyR};
    double cs[4], fx[4], fy[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const double va = tanh_e(10.0 * v_lon[w]) * v_lon[w] + 1e-6 * exp(-(v_lon[w] * v_lon[w]));
        const double cl = lat_pacejka_t(atan2(v_lat[w], va));
        const double bs = Bs * (x[6 + w] * k_Rw / va - 1.0);
        cs[w] = Ds * sin(Cs * atan(bs - Es * (bs - atan(bs))));
        // body-frame force per unit of normal load N_w = -F_z,w:  F_lon = N cs, F_lat = -N cl
        if (w < 2) { fx[w] = cd * cs[w] + sd * cl; fy[w] = sd * cs[w] - cd * cl; }
        else { fx[w] = cs[w]; fy[w] = -cl; }
    }
    // N_w = base + sx_w cx a_x + sy_w cy a_y, sx = (-,-,+,+), sy = (+,-,+,-)
    const double Sfx = fx[0] + fx[1] + fx[2] + fx[3], Sfy = fy[0] + fy[1] + fy[2] + fy[3];
    const double Sxx = -fx[0] - fx[1] + fx[2] + fx[3], Sxy = fx[0] - fx[1] + fx[2] - fx[3];
    const double Syx = -fy[0] - fy[1] + fy[2] + fy[3], Syy = fy[0] - fy[1] + fy[2] - fy[3];
    const double a11 = k_m - cx * Sxx, a12 = -cy * Sxy, a21 = -cx * Syx, a22 = k_m - cy * Syy;
    const double b1 = F_drag + base * Sfx, b2 = base * Sfy, det = a11 * a22 - a12 * a21;
    const double a_x = (b1 * a22 - a12 * b2) / det, a_y = (a11 * b2 - a21 * b1) / det;
    const double lx = cx * a_x, ly = cy * a_y;
    const double Nw[4] = {base - lx + ly, base - lx - ly, base + lx + ly, base + lx - ly};
    double dk;
    const double kap = trk.kappa(x[0], dk);
    double sp, cp;
    fast_sincos(psi, &sp, &cp);
    const double s_dot = (v_x * cp - v_y * sp) / (1.0 + kap * n);
    f[0] = s_dot;
    f[1] = v_x * sp + v_y * cp;
    f[2] = r - kap * s_dot;
    f[3] = a_x + v_y * r;
    f[4] = a_y - v_x * r;
    f[5] = ((Nw[1] * fx[1] - Nw[0] * fx[0]) * hx + (Nw[1] * fy[1] + Nw[0] * fy[0]) * k_lF + (Nw[3] * fx[3] - Nw[2] * fx[2]) * hx
            - (Nw[3] * fy[3] + Nw[2] * fy[2]) * k_lR) * (1.0 / k_Iz);
#pragma unroll
    for (int w = 0; w < 4; w++) {
        f[6 + w] = (x[10 + w] - (k_kd * x[6 + w] + k_ks + k_Rw * (Nw[w] * cs[w]))) * (1.0 / k_Iw);
        f[10 + w] = (u[w] - x[10 + w]) * (1.0 / k_tT);
    }
    f[14] = (u[4] - delta) * (1.0 / k_tdelta);
}

// plant step of the 15-state model: RK4 x M over dt; one lane per instance
__global__ __launch_bounds__(64) void k_sim_dyn10(int B, int M, double dt, int nknots, const double *__restrict__ s_ref, const double *__restrict__ kappa_ref,
                                                  const int32_t *__restrict__ track_id, const double *xs, const double *__restrict__ us, double *xn)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double x[15], u[5];
#pragma unroll
    for (int i = 0; i < 15; i++) x[i] = xs[(size_t)b * 15 + i];
#pragma unroll
    for (int i = 0; i < 5; i++) u[i] = us[(size_t)b * 5 + i];
    const int tid = track_id[b];
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, x[0]);
    const double h = dt / M;
    for (int m = 0; m < M; m++) {
        double xacc[15], K[15];
#pragma unroll
        for (int i = 0; i < 15; i++) { xacc[i] = x[i]; K[i] = 0.0; }
#pragma unroll 1
        for (int st = 0; st < 4; st++) {
            const double ah = (st == 0) ? 0.0 : ((st == 3) ? h : 0.5 * h);
            const double wh = (st == 0 || st == 3) ? h * (1.0 / 6.0) : h * (2.0 / 6.0);
            double X[15];
#pragma unroll
            for (int i = 0; i < 15; i++) X[i] = fma(ah, K[i], x[i]);
            fdyn10_rhs(X, u, trk, K);
#pragma unroll
            for (int i = 0; i < 15; i++) xacc[i] = fma(wh, K[i], xacc[i]);
        }
#pragma unroll
        for (int i = 0; i < 15; i++) x[i] = xacc[i];
    }
#pragma unroll
    for (int i = 0; i < 15; i++) xn[(size_t)b * 15 + i] = x[i];
}

// ---- closed cubic-spline fit of a centre line (python/motion_planning.py:28-124), one workgroup per track ----
// The reference hands  min 1/2 p'P p + q'p  s.t.  A p = 0  (P = B'B + w C'C + 1e-10 I: interpolation error + curvature, :88-101; A: value, first
// and second derivative continuity between consecutive segments of the CLOSED path, :59-87) to qpsolvers / proxqp.  It is one linear system:
// the KKT matrix [[P, A'], [A, 0]] (7n x 7n for n points, cyclic-banded) with the right-hand sides (-q_x, 0), (-q_y, 0).  Solved here by
// Gaussian elimination with partial pivoting on the augmented matrix in global memory (row-major, m + 2 columns): per pivot the rows and the
// columns that are non-zero are listed in LDS first -- the matrix stays sparse under elimination, so a step touches tens of entries, not m^2.
constexpr int FIT_T = 256, FIT_LIST = 1280;
__global__ __launch_bounds__(FIT_T) void k_track_fit(int max_pts, const int32_t *__restrict__ npts, const double *__restrict__ xy, double curv_weight,
                                                     double *__restrict__ work, double *__restrict__ cX, double *__restrict__ cY, int32_t *__restrict__ fail_flag)
{
    __shared__ int rows[FIT_LIST], cols[FIT_LIST], n_rows, n_cols;
    __shared__ double red_v[FIT_T];
    __shared__ int red_i[FIT_T];
    const int trk = blockIdx.x, tid = threadIdx.x;
    const int n = npts[trk], m = 7 * n, ld = m + 2;
    const double *pts = xy + (size_t)trk * max_pts * 2;
    double *Mx = work + (size_t)trk * (7 * (size_t)max_pts) * (7 * (size_t)max_pts + 2);
    for (size_t e = tid; e < (size_t)m * ld; e += FIT_T) Mx[e] = 0.0;
    __syncthreads();
    for (int i = tid; i < n; i += FIT_T) {
        const int nx = (i + 1) % n, pv = (i + n - 1) % n;
        const double dsi = hypot(pts[nx * 2] - pts[i * 2], pts[nx * 2 + 1] - pts[i * 2 + 1]);             // chord i -> i + 1 (closing chord last)
        const double dsn = hypot(pts[((nx + 1) % n) * 2] - pts[nx * 2], pts[((nx + 1) % n) * 2 + 1] - pts[nx * 2 + 1]);
        const double rho = dsi / dsn;
        // P block of segment i: B'B on c0, w C'C on (c2, c3), 1e-10 on the diagonal
        const double w2 = 2.0 / (dsi * dsi), w3 = 6.0 / (dsi * dsi);
        double *Pi = Mx + (size_t)(4 * i) * ld + 4 * i;
        Pi[0] = 1.0 + 1e-10; Pi[ld + 1] = 1e-10;
        Pi[2 * ld + 2] = curv_weight * w2 * w2 + 1e-10; Pi[2 * ld + 3] = curv_weight * w2 * w3;
        Pi[3 * ld + 2] = curv_weight * w3 * w2;          Pi[3 * ld + 3] = curv_weight * w3 * w3 + 1e-10;
        // continuity rows 3i .. 3i + 2 of A (and their transposes): segment i at t = 1 against segment i + 1 at t = 0
        const double a[3][4] = {{1, 1, 1, 1}, {0, 1, 2, 3}, {0, 0, 2, 6}};
        const double nb[3] = {-1.0, -rho, -2.0 * rho * rho};
        for (int q = 0; q < 3; q++) {
            const int r = 4 * n + 3 * i + q;
            for (int c = 0; c < 4; c++)
                if (a[q][c] != 0.0) { Mx[(size_t)r * ld + 4 * i + c] = a[q][c]; Mx[(size_t)(4 * i + c) * ld + r] = a[q][c]; }
            // (n = 2 would put both junctions of a segment on the same coefficients: accumulate)
            Mx[(size_t)r * ld + 4 * nx + q] += nb[q]; Mx[(size_t)(4 * nx + q) * ld + r] += nb[q];
        }
        (void)pv;
        Mx[(size_t)(4 * i) * ld + m] = pts[i * 2];          // -q = B' path
        Mx[(size_t)(4 * i) * ld + m + 1] = pts[i * 2 + 1];
    }
    __syncthreads();
    bool singular = false;
    for (int c = 0; c < m; c++) {
        // pivot: the first row with the largest |entry| in column c
        double bv = -1.0; int bi = c;
        for (int r = c + tid; r < m; r += FIT_T) { const double v = fabs(Mx[(size_t)r * ld + c]); if (v > bv) { bv = v; bi = r; } }
        red_v[tid] = bv; red_i[tid] = bi;
        __syncthreads();
        for (int off = FIT_T / 2; off > 0; off >>= 1) {
            if (tid < off) {
                const double ov = red_v[tid + off]; const int oi = red_i[tid + off];
                if (ov > red_v[tid] || (ov == red_v[tid] && oi < red_i[tid])) { red_v[tid] = ov; red_i[tid] = oi; }
            }
            __syncthreads();
        }
        const int p = red_i[0];
        if (!(red_v[0] > 0.0)) { singular = true; break; }
        if (tid == 0) { n_rows = 0; n_cols = 0; }
        __syncthreads();
        if (p != c)
            for (int j = c + tid; j < ld; j += FIT_T) { const double t0 = Mx[(size_t)c * ld + j], t1 = Mx[(size_t)p * ld + j]; Mx[(size_t)c * ld + j] = t1; Mx[(size_t)p * ld + j] = t0; }
        __syncthreads();
        // non-zero columns of the pivot row (right of the pivot, right-hand sides included) and rows with a non-zero entry under the pivot
        for (int j = c + 1 + tid; j < ld; j += FIT_T) if (Mx[(size_t)c * ld + j] != 0.0) { const int q = atomicAdd(&n_cols, 1); if (q < FIT_LIST) cols[q] = j; }
        for (int r = c + 1 + tid; r < m; r += FIT_T) if (Mx[(size_t)r * ld + c] != 0.0) { const int q = atomicAdd(&n_rows, 1); if (q < FIT_LIST) rows[q] = r; }
        __syncthreads();
        const int nr = min(n_rows, FIT_LIST), nc = min(n_cols, FIT_LIST);      // (m + 2 <= FIT_LIST is checked on the host)
        const double ipiv = 1.0 / Mx[(size_t)c * ld + c];
        for (int e = tid; e < nr * nc; e += FIT_T) {
            const int r = rows[e / nc], j = cols[e % nc];
            Mx[(size_t)r * ld + j] -= Mx[(size_t)r * ld + c] * ipiv * Mx[(size_t)c * ld + j];
        }
        __syncthreads();
    }
    if (singular) { if (tid == 0) fail_flag[trk] = 1; return; }
    // back substitution, both right-hand sides: x_c = (rhs_c - sum_{j > c} M_cj x_j) / M_cc; the solution overwrites the right-hand sides
    for (int c = m - 1; c >= 0; c--) {
        double s0 = 0.0, s1 = 0.0;
        for (int j = c + 1 + tid; j < m; j += FIT_T) {
            const double v = Mx[(size_t)c * ld + j];
            if (v != 0.0) { s0 = fma(v, Mx[(size_t)j * ld + m], s0); s1 = fma(v, Mx[(size_t)j * ld + m + 1], s1); }
        }
        red_v[tid] = s0;
        __syncthreads();
        for (int off = FIT_T / 2; off > 0; off >>= 1) { if (tid < off) red_v[tid] += red_v[tid + off]; __syncthreads(); }
        const double t0 = red_v[0];
        __syncthreads();
        red_v[tid] = s1;
        __syncthreads();
        for (int off = FIT_T / 2; off > 0; off >>= 1) { if (tid < off) red_v[tid] += red_v[tid + off]; __syncthreads(); }
        if (tid == 0) {
            const double d = Mx[(size_t)c * ld + c];
            Mx[(size_t)c * ld + m] = (Mx[(size_t)c * ld + m] - t0) / d;
            Mx[(size_t)c * ld + m + 1] = (Mx[(size_t)c * ld + m + 1] - red_v[0]) / d;
        }
        __syncthreads();
    }
    for (int e = tid; e < 4 * n; e += FIT_T) {
        cX[(size_t)trk * max_pts * 4 + e] = Mx[(size_t)e * ld + m];
        cY[(size_t)trk * max_pts * 4 + e] = Mx[(size_t)e * ld + m + 1];
    }
    if (tid == 0) fail_flag[trk] = 0;
}

// n_steps plant steps of length dt (each RK4 x M); model 3 = kin6, 4 = dyn6, -3 = the node's speed switch + no reversing
__global__ __launch_bounds__(64) void k_sim_cart(int B, int model, int M, double dt, int n_steps, double v_dyn, const double *xs,
                                                 const double *__restrict__ us, double *xn)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = xs[(size_t)b * 8 + i];
    const double u_T = us[(size_t)b * 2], u_d = us[(size_t)b * 2 + 1];
    const double h = dt / M;
    for (int step = 0; step < n_steps; step++) {
        int mdl = model;
        if (model == -3) mdl = (hypot(x[3], x[4]) < v_dyn) ? IHM2MPC_PLANT_KIN6 : IHM2MPC_PLANT_DYN6;
        for (int m = 0; m < M; m++) {
            double xacc[8], K[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { xacc[i] = x[i]; K[i] = 0.0; }
#pragma unroll 1
            for (int st = 0; st < 4; st++) {
                const double ah = (st == 0) ? 0.0 : ((st == 3) ? h : 0.5 * h);
                const double wh = (st == 0 || st == 3) ? h * (1.0 / 6.0) : h * (2.0 / 6.0);
                double X[8];
#pragma unroll
                for (int i = 0; i < 8; i++) X[i] = fma(ah, K[i], x[i]);
                if (mdl == IHM2MPC_PLANT_KIN6) kin6_rhs(X, u_T, u_d, K);
                else dyn6_rhs(X, u_T, u_d, K);
#pragma unroll
                for (int i = 0; i < 8; i++) xacc[i] = fma(wh, K[i], xacc[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = xacc[i];
        }
        if (model == -3 && (x[3] < 0.0 || (x[6] <= 0.1 && x[3] < 0.01))) { x[3] = 0.0; x[4] = 0.0; x[5] = 0.0; }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) xn[(size_t)b * 8 + i] = x[i];
}

__device__ __forceinline__ double wrap_to_pi(double x)
{
    double t = fmod(x + M_PI, 2.0 * M_PI);
    if (t < 0.0) t += 2.0 * M_PI;
    return t - M_PI;
}
// index of the last element <= x, -1 if x < v[0]
__device__ inline int locate_index(const double *v, int n, double x)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}
__device__ __forceinline__ double angle3pt(double ax, double ay, double bx, double by, double cx, double cy)
{
    return wrap_to_pi(atan2(cy - by, cx - bx) - atan2(ay - by, ax - bx));
}

__global__ __launch_bounds__(64) void k_project(int B, int nk, double s_tol, const double *__restrict__ s_ref,
                                                const double *__restrict__ X_ref, const double *__restrict__ Y_ref,
                                                const double *__restrict__ phi_ref, const int32_t *__restrict__ track_id,
                                                const double *__restrict__ xc, double *s_guess, double *xf)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int t = track_id[b];
    const double *sr = s_ref + (size_t)t * nk, *Xr = X_ref + (size_t)t * nk, *Yr = Y_ref + (size_t)t * nk, *pr = phi_ref + (size_t)t * nk;
    const double X = xc[(size_t)b * 8], Y = xc[(size_t)b * 8 + 1], sg = s_guess[b];
    const double s_low = fmax(sg - s_tol, sr[0]), s_up = fmin(sg + s_tol, sr[nk - 1]);
    int id_low = locate_index(sr, nk, s_low), id_up = locate_index(sr, nk, s_up);
    if (id_low > 0) --id_low;
    if (id_up < nk - 1) ++id_up;
    const int nloc = id_up - id_low + 1;
    int id_min = 0;
    double best = INFINITY;
    for (int i = 0; i < nloc; i++) {
        const double dx = Xr[id_low + i] - X, dy = Yr[id_low + i] - Y, d = dx * dx + dy * dy;
        if (d < best) { best = d; id_min = i; }
    }
    const int id_prev = (id_min == 0) ? nloc - 1 : id_min - 1;
    const int id_next = (id_min == nloc - 1) ? 0 : id_min + 1;
    const double mx = Xr[id_low + id_min], my = Yr[id_low + id_min];
    const double px = Xr[id_low + id_prev], py = Yr[id_low + id_prev];
    const double nx = Xr[id_low + id_next], ny = Yr[id_low + id_next];
    const double angle_prev = fabs(angle3pt(mx, my, X, Y, px, py)), angle_next = fabs(angle3pt(mx, my, X, Y, nx, ny));
    const bool use_prev = angle_prev > angle_next;
    const double ax = use_prev ? px : mx, ay = use_prev ? py : my, bx = use_prev ? mx : nx, by = use_prev ? my : ny;
    const double sa = use_prev ? sr[id_prev + id_low] : sr[id_min + id_low], sb = use_prev ? sr[id_min + id_low] : sr[id_next + id_low];
    const double dx = bx - ax, dy = by - ay;
    const double lambda = ((X - ax) * dx + (Y - ay) * dy) / (dx * dx + dy * dy);
    const double s = sa + lambda * (sb - sa), Xp = ax + lambda * dx, Yp = ay + lambda * dy;
    int ind = id_min + id_low;
    if (ind > nk - 2) ind = nk - 2;
    const double phi_p = pr[ind] + (pr[ind + 1] - pr[ind]) / (sr[ind + 1] - sr[ind]) * (s - sr[ind]);
    const double rho = wrap_to_pi(phi_p);
    const double psi = wrap_to_pi(wrap_to_pi(xc[(size_t)b * 8 + 2]) - rho);
    const double e = hypot(Xp - X, Yp - Y);
    double sr_, cr_;
    sincos(rho, &sr_, &cr_);
    const double tpr = (Y - Yp) * cr_ - (X - Xp) * sr_;
    double *o = xf + (size_t)b * 8;
    o[0] = s;
    o[1] = e * (tpr > 0.0 ? 1.0 : -1.0);
    o[2] = psi;
#pragma unroll
    for (int i = 3; i < 8; i++) o[i] = xc[(size_t)b * 8 + i];
    s_guess[b] = fmod(s + xc[(size_t)b * 8 + 3] * 0.05, -sr[0]);
}

// ---- track tables on the device (python/motion_planning.py:139-289, 345-428) ----
// In: the cubic spline coefficients of every segment of every centre line (the closed-spline fit itself is one small KKT solve per
// track on the host).  Out: the tables the models and the projection read -- s_ref, kappa_ref, X_ref, Y_ref, phi_ref on nknots = 3 x
// n_samples knots (three laps side by side).
__device__ __forceinline__ double poly3(const double *c, double t, int der)
{
    if (der == 0) return c[0] + t * (c[1] + t * (c[2] + t * c[3]));
    if (der == 1) return c[1] + t * (2.0 * c[2] + 3.0 * c[3] * t);
    return 2.0 * c[2] + 6.0 * c[3] * t;
}
// polyline length of a segment on 100 points (compute_spline_interval_lengths): one lane per (track, segment)
__global__ __launch_bounds__(64) void k_track_seglen(int ntracks, int max_seg, const int32_t *__restrict__ nseg, const double *__restrict__ cX,
                                                     const double *__restrict__ cY, double *seglen)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= ntracks * max_seg) return;
    const int t = e / max_seg, j = e - t * max_seg;
    double len = 0.0;
    if (j < nseg[t]) {
        const double *cx = cX + (size_t)e * 4, *cy = cY + (size_t)e * 4;
        double xp = poly3(cx, 0.0, 0), yp = poly3(cy, 0.0, 0);
        for (int i = 1; i < 100; i++) {
            const double tt = (i == 99) ? 1.0 : i * (1.0 / 99.0);
            const double xq = poly3(cx, tt, 0), yq = poly3(cy, tt, 0);
            len += hypot(xq - xp, yq - yp);
            xp = xq; yp = yq;
        }
    }
    seglen[e] = len;
}
// cumulative arc length at the segment ends, in place: one lane per track (a few dozen segments)
__global__ void k_track_cumsum(int ntracks, int max_seg, const int32_t *__restrict__ nseg, double *seglen_to_send)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntracks) return;
    double acc = 0.0;
    for (int j = 0; j < nseg[t]; j++) { acc += seglen_to_send[(size_t)t * max_seg + j]; seglen_to_send[(size_t)t * max_seg + j] = acc; }
}
// uniformly_sample_spline + get_heading + get_curvature + offline_motion_plan's heading offset + triple_motion_plan_ref:
// one lane per (track, sample); every lane writes its sample into the three laps
__global__ __launch_bounds__(64) void k_track_sample(int ntracks, int max_seg, int n_samples, const int32_t *__restrict__ nseg, const double *__restrict__ cX,
                                                     const double *__restrict__ cY, const double *__restrict__ s_end, double *s_ref, double *kappa_ref,
                                                     double *X_ref, double *Y_ref, double *phi_ref)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= ntracks * n_samples) return;
    const int t = e / n_samples, i = e - t * n_samples;
    const int ns = nseg[t];
    const double *se = s_end + (size_t)t * max_seg;
    const double total = se[ns - 1], step = total / n_samples;
    auto sample = [&](int k, double &X, double &Y, double &phi, double &kap) -> double {
        const double s = k * step;
        int idx = 0;
        while (idx < ns - 1 && !(s < se[idx])) idx++;           // first segment whose end lies beyond s
        const double s0 = (idx == 0) ? 0.0 : se[idx - 1];
        const double tt = (s - s0) / (se[idx] - s0);
        const double *cx = cX + ((size_t)t * max_seg + idx) * 4, *cy = cY + ((size_t)t * max_seg + idx) * 4;
        X = poly3(cx, tt, 0); Y = poly3(cy, tt, 0);
        const double xd = poly3(cx, tt, 1), yd = poly3(cy, tt, 1), xdd = poly3(cx, tt, 2), ydd = poly3(cy, tt, 2);
        const double q = xd * xd + yd * yd;
        kap = (xd * ydd - yd * xdd) / (q * sqrt(q));
        phi = atan2(yd, xd) - asin(k_lR * kap);
        return s;
    };
    double X, Y, phi, kap, Xl, Yl, pl, kl, X0, Y0, p0, k0;
    const double s = sample(i, X, Y, phi, kap);
    const double s_last = sample(n_samples - 1, Xl, Yl, pl, kl);
    (void)sample(0, X0, Y0, p0, k0);
    const double L = s_last + hypot(Xl - X0, Yl - Y0);         // lap length as offline_motion_plan closes it
    const int nk = 3 * n_samples;
    for (int lap = 0; lap < 3; lap++) {
        const size_t o = (size_t)t * nk + (size_t)lap * n_samples + i;
        s_ref[o] = s + (lap - 1) * L; kappa_ref[o] = kap; X_ref[o] = X; Y_ref[o] = Y; phi_ref[o] = phi;
    }
}

}  // namespace

void ihm2_launch_build_tracks(ihm2mpc_handle *h, int max_seg, const int32_t *nseg, const double *cX, const double *cY, double *work)
{
    const int nt = h->cfg.ntracks, n_samples = h->cfg.nknots / 3;
    hipLaunchKernelGGL(k_track_seglen, dim3((nt * max_seg + 63) / 64), dim3(64), 0, h->stream, nt, max_seg, nseg, cX, cY, work);
    hipLaunchKernelGGL(k_track_cumsum, dim3((nt + 63) / 64), dim3(64), 0, h->stream, nt, max_seg, nseg, work);
    hipLaunchKernelGGL(k_track_sample, dim3((nt * n_samples + 63) / 64), dim3(64), 0, h->stream, nt, max_seg, n_samples, nseg, cX, cY, work,
                       h->s_ref, h->kappa_ref, h->X_ref, h->Y_ref, h->phi_ref);
}

void ihm2_launch_sim_cart(ihm2mpc_handle *h, int model, int M, double dt, int n_steps, double v_dyn, const double *x, const double *u,
                          double *xn, hipStream_t stream)
{
    hipLaunchKernelGGL(k_sim_cart, dim3((h->B + 63) / 64), dim3(64), 0, stream, h->B, model, M, dt, n_steps, v_dyn, x, u, xn);
}

void ihm2_launch_sim_dyn10(ihm2mpc_handle *h, int M, const double *x, const double *u, double *xn, hipStream_t stream)
{
    hipLaunchKernelGGL(k_sim_dyn10, dim3((h->B + 63) / 64), dim3(64), 0, stream, h->B, M, h->cfg.dt, h->cfg.nknots, h->s_ref, h->kappa_ref, h->track_id, x, u, xn);
}

void ihm2_launch_project(ihm2mpc_handle *h, double s_tol, const double *xc, double *s_guess, double *xf, hipStream_t stream)
{
    hipLaunchKernelGGL(k_project, dim3((h->B + 63) / 64), dim3(64), 0, stream, h->B, h->cfg.nknots, s_tol, h->s_ref, h->X_ref, h->Y_ref,
                       h->phi_ref, h->track_id, xc, s_guess, xf);
}

// closed-spline fit of every track's centre line on the device: coefficients (ntracks, max_pts, 4) in cX, cY (device); work: ntracks x 7 max_pts x (7 max_pts + 2)
void ihm2_launch_track_fit(ihm2mpc_handle *h, int max_pts, const int32_t *npts, const double *xy, double curv_weight, double *work, double *cX, double *cY,
                           int32_t *fail_flag)
{
    hipLaunchKernelGGL(k_track_fit, dim3(h->cfg.ntracks), dim3(FIT_T), 0, h->stream, max_pts, npts, xy, curv_weight, work, cX, cY, fail_flag);
}
