// sqp_body.hpp -- the convergence test + merit line search of the SQP mode as a device function (see kernels_sqp.hip for the
// algorithm), shared by k_line_search and the persistent per-instance loop (kernels_qp.hip, k_steps).
#pragma once

#include "ihm2mpc_internal.h"
#include "model.hpp"
#include "irk_body.hpp"

namespace ihm2 {


struct LsArgs {
    int B, N, M, nknots, globalization, use_suff, full_step_dual, path_on;
    double dt, cs, alpha_min, alpha_red, eps, car_L, car_W;
    double tol[4];
    const double *s_ref, *kappa_ref;
    const int32_t *track_id;
    const double *W;                 // (N,12,12) then W_e (8,8)
    const double *st_lb, *st_ub;     // (NS,14) bounds per (stage, row), -+inf = absent
    const double *st_sz, *st_sZ;     // (NS,28) slack penalties per side, sZ < 0 = hard
    const double *CD, *Hs, *widths;
    const double *x0, *yref, *yref_e, *g, *lin;
    double *x, *u, *pi, *lam, *slk;  // in: the QP's full step; out: the accepted iterate
    const double *xp, *up, *pip, *lamp, *slkp;   // the iterate the QP was built at
    double *wpi, *wlam;
    const double *res;
    int32_t *status, *qp_iter, *done, *sqp_status, *sqp_iter, *qp_acc;
    double *alpha, *u0;
    const double *phi;               // IRK: Phi at the trial points, (n_alpha, B, N, 8), from k_rollout_irk; nullptr: RK4 rollouts in this kernel
    int n_alpha;
    // the ladder in two launches (IRK): phase 1 tries the first j_limit step lengths and marks the instances that need more in
    // `pending` without touching their iterate; phase 2 redoes the ladder of exactly those (same arithmetic, all rollouts present);
    // phase 0: one launch, the whole ladder; phase 3 (persistent loop): the rollouts are done here, before each trial
    int phase, j_limit;
    int32_t *pending;
    const IrkTab *irk_tab;
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int MODEL>
__device__ __forceinline__ void rollout(double (&x)[8], double u_T, double u_d, TrackSeg &trk, int M, double h)
{
    double J[8][10];
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
            if (MODEL == IHM2MPC_MODEL_FKIN6) fkin6_eval<false>(X, u_T, u_d, trk, K, J);
            else fdyn6_eval<false, MODEL == IHM2MPC_MODEL_FDYN6U>(X, u_T, u_d, trk, K, J);
#pragma unroll
            for (int i = 0; i < 8; i++) xacc[i] = fma(wh, K[i], xacc[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = xacc[i];
    }
}

// The convergence test + line search of instance b, run by the calling wavefront (lane = threadIdx.x).  it: SQP iteration index
// of this solve, last: it == max_iter - 1.  Called by k_line_search (one launch per iteration) and by the persistent loop.
// ROLL: the collocation rollouts of the trial points are done here, one step length at a time (persistent loop with IRK; LsArgs.phase = 3)
template <int MODEL, bool ROLL = false>
__device__ __forceinline__ void line_search_body(const LsArgs &a, const int b, const int it, const int last)
{
    const int lane = threadIdx.x;
    const int N = a.N, NS = N + 1;
    double *xb = a.x + (size_t)b * NS * 8, *ub = a.u + (size_t)b * N * 2, *pib = a.pi + (size_t)b * NS * 8;
    double *lamb = a.lam + (size_t)b * NS * 28, *slb = a.slk + (size_t)b * NS * 28;
    const double *xpb = a.xp + (size_t)b * NS * 8, *upb = a.up + (size_t)b * N * 2, *pipb = a.pip + (size_t)b * NS * 8;
    const double *lampb = a.lamp + (size_t)b * NS * 28, *slpb = a.slkp + (size_t)b * NS * 28;
    double *wpib = a.wpi + (size_t)b * NS * 8, *wlamb = a.wlam + (size_t)b * NS * 28;
    const double *x0b = a.x0 + (size_t)b * 8;

    auto restore = [&](bool primal_dual) {
        if (primal_dual) {
            for (int e = lane; e < NS * 8; e += 64) { xb[e] = xpb[e]; pib[e] = pipb[e]; }
            for (int e = lane; e < N * 2; e += 64) ub[e] = upb[e];
            for (int e = lane; e < NS * 28; e += 64) lamb[e] = lampb[e];
        }
        for (int e = lane; e < NS * 28; e += 64) slb[e] = slpb[e];
        if (lane < 2) a.u0[(size_t)b * 2 + lane] = upb[lane];
    };
    auto finish = [&](int st_now) {      // what get_status / get_qp_iter report after this iteration
        if (lane == 0) { a.status[b] = st_now; a.qp_iter[b] = a.qp_acc[b]; }
    };

    // everything below is wave-uniform
    const bool resume = a.phase == 2;       // the bookkeeping and the merit weights of this iteration were done by phase 1
    if (resume && !a.pending[b]) return;
    if (a.phase == 1 && lane == 0) a.pending[b] = 0;
    if (!resume) {
    if (a.done[b]) { restore(true); finish(a.sqp_status[b]); return; }
    const int qst = a.status[b];
    if (lane == 0) a.qp_acc[b] += a.qp_iter[b];
    const double *rs = a.res + (size_t)b * 4;
    if (rs[0] <= a.tol[0] && rs[1] <= a.tol[1] && rs[2] <= a.tol[2] && rs[3] <= a.tol[3]) {
        restore(true);
        if (lane == 0) { a.done[b] = 1; a.sqp_status[b] = 0; if (it == 0) a.alpha[b] = 1.0; }   // alpha: last step taken, 1 if none
        __syncthreads();
        finish(0);
        return;
    }
    if (lane == 0) a.sqp_iter[b] += 1;
    if (qst != 0) {                      // the QP kernel left x, u, pi, lam as they were
        restore(false);
        if (lane == 0) { a.done[b] = 1; a.sqp_status[b] = qst; if (it == 0) a.alpha[b] = 1.0; }
        __syncthreads();
        finish(qst);
        return;
    }
    if (!a.globalization) {
        if (lane == 0) a.alpha[b] = 1.0;
        __syncthreads();
        finish(last ? 2 : 0);
        return;
    }

    // ---- merit weights (pi_0: multiplier of x_0 = x0, from the stage-0 stationarity row of the QP) ----
    {
        double pi0 = 0.0;
        if (lane < 8) {
            const int j = lane;
            const double *gb = a.g + (size_t)b * NS * 10, *rec = a.lin + (size_t)b * N * LIN_REC;
            pi0 = gb[j];
            for (int l = 0; l < 10; l++) pi0 = fma(a.Hs[j * 10 + l], (l < 8) ? xb[l] - xpb[l] : ub[l - 8] - upb[l - 8], pi0);
            for (int l = 0; l < 8; l++) pi0 = fma(rec[l * 8 + j], pib[8 + l], pi0);
            for (int r = 0; r < 2; r++) pi0 = fma(-a.CD[r * 10 + j], lamb[10 + r] - lamb[24 + r], pi0);
        }
        for (int e = lane; e < NS * 8; e += 64) {
            const double m = fabs(e < 8 ? pi0 : pib[e]);
            wpib[e] = (it == 0) ? m : fmax(m, 0.5 * (wpib[e] + m));
        }
        for (int e = lane; e < NS * 28; e += 64) {
            const double m = fabs(lamb[e]);
            wlamb[e] = (it == 0) ? m : fmax(m, 0.5 * (wlamb[e] + m));
        }
    }
    }
    __syncthreads();

    const int tid = a.track_id[b];
    const double *sr = a.s_ref + (size_t)tid * a.nknots, *kr = a.kappa_ref + (size_t)tid * a.nknots;
    const double w_R = a.path_on ? a.widths[tid * 2 + 0] : 0.0, w_L = a.path_on ? a.widths[tid * 2 + 1] : 0.0;
    const double hstep = a.dt / a.M;

    // merit at xp + al (x - xp): cost (with the slack penalties) and weighted infeasibility, summed over the wave;
    // at_lin: al = 0, the dynamics defects are the b_k of the linearisation records (no rollout)
    const double *linb = a.lin + (size_t)b * N * LIN_REC;
    auto merit = [&](double al, int jtrial, bool at_lin, double &cost_out, double &inf_out) {
        double cost = 0.0, inf = 0.0;
        for (int k = lane; k < NS; k += 64) {
            double xk[8], cv[14];
#pragma unroll
            for (int i = 0; i < 8; i++) { xk[i] = xpb[k * 8 + i] + al * (xb[k * 8 + i] - xpb[k * 8 + i]); cv[i] = xk[i]; }
#pragma unroll
            for (int i = 8; i < 14; i++) cv[i] = 0.0;
            if (k == 0)
                for (int i = 0; i < 8; i++) inf += wpib[i] * fabs(x0b[i] - xk[i]);
            if (k < N) {
                const double uT = upb[k * 2] + al * (ub[k * 2] - upb[k * 2]), ud = upb[k * 2 + 1] + al * (ub[k * 2 + 1] - upb[k * 2 + 1]);
                const double *yr = a.yref + ((size_t)b * N + k) * 12, *Wk = a.W + (size_t)k * 144;
                double e[12];
#pragma unroll
                for (int i = 0; i < 8; i++) e[i] = xk[i] - yr[i];
                e[8] = uT - yr[8]; e[9] = ud - yr[9]; e[10] = xk[6] - uT - yr[10]; e[11] = xk[7] - ud - yr[11];
                double q = 0.0;
                for (int i = 0; i < 12; i++) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < 12; j++) acc = fma(Wk[i * 12 + j], e[j], acc);
                    q = fma(e[i], acc, q);
                }
                cost += 0.5 * a.cs * q;
                if (at_lin) {
                    for (int i = 0; i < 8; i++) inf += wpib[(k + 1) * 8 + i] * fabs(linb[(size_t)k * LIN_REC + 80 + i]);
                } else {
                    double xn[8];
                    if (a.phi) {            // collocation integrator: the rollout of this trial point was done by k_rollout_irk
                        const double *ph = a.phi + (((size_t)jtrial * a.B + b) * N + k) * 8;
#pragma unroll
                        for (int i = 0; i < 8; i++) xn[i] = ph[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; i++) xn[i] = xk[i];
                        TrackSeg trk;
                        trk.init(sr, kr, a.nknots, xn[0]);
                        rollout<MODEL>(xn, uT, ud, trk, a.M, hstep);
                    }
                    for (int i = 0; i < 8; i++) {
                        const int e1 = (k + 1) * 8 + i;
                        inf += wpib[e1] * fabs(xn[i] - (xpb[e1] + al * (xb[e1] - xpb[e1])));
                    }
                }
                cv[8] = uT; cv[9] = ud;
                for (int r = 0; r < 2; r++) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < 8; j++) acc = fma(a.CD[(k * 2 + r) * 10 + j], xk[j], acc);
                    acc = fma(a.CD[(k * 2 + r) * 10 + 8], uT, acc);
                    acc = fma(a.CD[(k * 2 + r) * 10 + 9], ud, acc);
                    cv[10 + r] = acc;
                }
            } else {
                const double *ye = a.yref_e + (size_t)b * 8, *We = a.W + (size_t)N * 144;
                double q = 0.0;
                for (int i = 0; i < 8; i++) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < 8; j++) acc = fma(We[i * 8 + j], xk[j] - ye[j], acc);
                    q = fma(xk[i] - ye[i], acc, q);
                }
                cost += 0.5 * q;
            }
            if (a.path_on && k >= 1) {
                const double foot = -0.5 * a.car_L * sin(fabs(xk[2])), lat = 0.5 * a.car_W * cos(xk[2]);
                cv[12] = xk[1] + foot + lat - w_R;
                cv[13] = -xk[1] - foot + lat - w_L;
            }
#pragma unroll
            for (int c = 0; c < 14; c++) {
                const double lb = a.st_lb[k * 14 + c], ubd = a.st_ub[k * 14 + c];
                if (lb > -INFINITY) {
                    double viol = lb - cv[c];
                    const double Z = a.st_sZ[k * 28 + c];
                    if (Z >= 0.0) {
                        const int e1 = k * 28 + c;
                        const double sv = slpb[e1] + al * (slb[e1] - slpb[e1]);
                        cost += a.st_sz[e1] * sv + 0.5 * Z * sv * sv;
                        viol -= sv;
                    }
                    inf += wlamb[k * 28 + c] * fmax(0.0, viol);
                }
                if (ubd < INFINITY) {
                    double viol = cv[c] - ubd;
                    const double Z = a.st_sZ[k * 28 + 14 + c];
                    if (Z >= 0.0) {
                        const int e1 = k * 28 + 14 + c;
                        const double sv = slpb[e1] + al * (slb[e1] - slpb[e1]);
                        cost += a.st_sz[e1] * sv + 0.5 * Z * sv * sv;
                        viol -= sv;
                    }
                    inf += wlamb[k * 28 + 14 + c] * fmax(0.0, viol);
                }
            }
        }
        cost_out = wave_sum(cost); inf_out = wave_sum(inf);
    };

    double c0, i0;
    merit(0.0, 0, true, c0, i0);
    const double m0 = c0 + i0;
    double D = 0.0;
    if (a.use_suff) {       // grad cost . step - infeasibility at alpha = 0 (the linearised constraints hold at the full step)
        const double *gb = a.g + (size_t)b * NS * 10;
        double acc = 0.0;
        for (int e = lane; e < NS * 10; e += 64) {
            const int k = e / 10, j = e % 10;
            if (j < 8) acc = fma(gb[e], xb[k * 8 + j] - xpb[k * 8 + j], acc);
            else if (k < N) acc = fma(gb[e], ub[k * 2 + j - 8] - upb[k * 2 + j - 8], acc);
        }
        for (int e = lane; e < NS * 28; e += 64)
            if (a.st_sZ[e] >= 0.0) acc = fma(a.st_sz[e] + a.st_sZ[e] * slpb[e], slb[e] - slpb[e], acc);
        D = fmin(wave_sum(acc) - i0, 0.0);
    }
    double al = 1.0;
    for (int jtrial = 0;; jtrial++) {              // at most log(alpha_min) / log(alpha_red) + 1 trials: al shrinks every pass
        if (a.phase == 1 && jtrial >= a.j_limit) {      // rollouts beyond this one are not there yet: phase 2 takes this instance
            if (lane == 0) a.pending[b] = 1;
            return;
        }
        if (ROLL) {         // collocation rollouts of this trial point: the wave's 16 quads over the intervals, then all lanes see them
            const int st = lane & 3;
            const IrkRows rows = irk_rows_from(a.irk_tab, st);
            for (int base = 0; base < N; base += 16) {
                const int q = base + (lane >> 2), k = min(q, N - 1);
                const size_t ex = ((size_t)b * NS + k) * 8, eu = ((size_t)b * N + k) * 2;
                irk_rollout_quad<MODEL>(st, rows, a.M, al, a.x + ex, a.xp + ex, a.u + eu, a.up + eu, tid, a.nknots, a.s_ref, a.kappa_ref,
                                        const_cast<double *>(a.phi) + (((size_t)jtrial * a.B + b) * N + k) * 8, q < N);
            }
            __syncthreads();
        }
        double c1, i1;
        merit(al, jtrial, false, c1, i1);
        const double m1 = c1 + i1;
        if (a.use_suff ? (m1 - m0 <= a.eps * al * D) : (m1 < m0)) break;
        al *= a.alpha_red;
        // (the second condition never decides: ihm2mpc_set_sqp_options refuses ladders longer than the rollout buffer; it keeps a
        // stale option block from reading or writing past phi)
        if (al < a.alpha_min || (a.phi && jtrial + 1 >= a.n_alpha)) { al = a.alpha_min; break; }
    }
    __syncthreads();
    for (int e = lane; e < NS * 8; e += 64) {
        xb[e] = xpb[e] + al * (xb[e] - xpb[e]);
        if (!a.full_step_dual) pib[e] = pipb[e] + al * (pib[e] - pipb[e]);
    }
    for (int e = lane; e < N * 2; e += 64) ub[e] = upb[e] + al * (ub[e] - upb[e]);
    for (int e = lane; e < NS * 28; e += 64) {
        slb[e] = slpb[e] + al * (slb[e] - slpb[e]);
        if (!a.full_step_dual) lamb[e] = lampb[e] + al * (lamb[e] - lampb[e]);
    }
    __syncthreads();
    if (lane < 2) a.u0[(size_t)b * 2 + lane] = ub[lane];
    if (lane == 0) a.alpha[b] = al;
    finish(last ? 2 : 0);
}


// kernel arguments from the handle
static inline LsArgs make_ls_args(ihm2mpc_handle *h)
{
    LsArgs a;
    a.B = h->B; a.N = h->N; a.M = h->cfg.M; a.nknots = h->cfg.nknots;
    a.globalization = h->sqp_globalization; a.use_suff = h->sqp_use_suff; a.full_step_dual = h->sqp_full_step_dual; a.path_on = h->path_on;
    a.dt = h->cfg.dt; a.cs = h->cfg.cost_scale_stage; a.alpha_min = h->sqp_alpha_min; a.alpha_red = h->sqp_alpha_red; a.eps = h->sqp_eps;
    a.car_L = h->car_L; a.car_W = h->car_W;
    for (int i = 0; i < 4; i++) a.tol[i] = h->sqp_tol[i];
    a.s_ref = h->s_ref; a.kappa_ref = h->kappa_ref; a.track_id = h->track_id;
    a.W = h->Wd; a.st_lb = h->st_lb; a.st_ub = h->st_ub; a.st_sz = h->st_sz; a.st_sZ = h->st_sZ;
    a.CD = h->CD; a.Hs = h->Hs; a.widths = h->widths;
    a.x0 = h->x0; a.yref = h->yref; a.yref_e = h->yref_e; a.g = h->q_g; a.lin = h->lin;
    a.x = h->x; a.u = h->u; a.pi = h->pi; a.lam = h->lam; a.slk = h->slk;
    a.xp = h->ls_x; a.up = h->ls_u; a.pip = h->ls_pi; a.lamp = h->ls_lam; a.slkp = h->ls_slk;
    a.wpi = h->ls_wpi; a.wlam = h->ls_wlam;
    a.res = h->res; a.status = h->status; a.qp_iter = h->qp_iter;
    a.done = h->ls_done; a.sqp_status = h->ls_status; a.sqp_iter = h->ls_iter; a.qp_acc = h->ls_qp_acc;
    a.alpha = h->ls_alpha; a.u0 = h->u0;
    a.phi = (h->cfg.integrator_type != IHM2MPC_INTEG_ERK) ? h->ls_phi : nullptr; a.n_alpha = h->ls_nalpha;
    a.phase = 0; a.j_limit = 0; a.pending = h->ls_pending; a.irk_tab = (const IrkTab *)h->irk_tab;
    return a;
}

}  // namespace ihm2
