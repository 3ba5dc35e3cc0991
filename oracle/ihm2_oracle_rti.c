/*
 * ihm2_oracle_rti.c -- CPU ORACLE (test infrastructure, NOT the product).
 * One SQP real-time iteration of the path-parametric bicycle NMPC for a batch of independent
 * instances (OpenMP over instances), plus the controller's warm-start shift.
 *
 * Follows (reference file:line, relative to /root/reference):
 *   OCP definition (cost selectors Vx/Vu, bounds, rate rows)  python/mpc.py:29-101
 *   weights, horizon, options                                  python/main.py:179-238
 *   per-step set/solve/get sequence                            python/main.py:297-334
 *   RTI semantics (acados, third party): SURVEY.md section 8c "RTI spec"
 *
 * NLP (multiple shooting, k = 0..N):
 *   min sum_{k<N} c_s/2 |Vx x_k + Vu u_k - yref_k|^2_W_k + 1/2 |x_N - yref_e|^2_W_e
 *   s.t. x_0 = x0, x_{k+1} = Phi(x_k,u_k), lbx<=x_k<=ubx (k>=1), lbu<=u_k<=ubu, lg<=C x_k+D u_k<=ug
 * One RTI iteration: linearise at (x,u), Gauss-Newton QP, full step, multipliers <- QP multipliers.
 */
#include "ihm2_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NX ORC_NX
#define NU ORC_NU
#define NZ ORC_NZ
#define NY ORC_NY
#define NC ORC_NC
#define INF_BOUND 1e20

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* y = Vx x + Vu u  (python/mpc.py:49-58):  y = [x(8); u(2); x[6]-u[0]; x[7]-u[1]] */
static void cost_selector(double V[NY * NZ])
{
    memset(V, 0, sizeof(double) * NY * NZ);
    for (int i = 0; i < NX; i++) V[i * NZ + i] = 1.0;       /* Vx[:nx] = I */
    V[10 * NZ + 6] = 1.0; V[11 * NZ + 7] = 1.0;              /* Vx[-nu:, -nu:] = I */
    V[8 * NZ + 8] = 1.0; V[9 * NZ + 9] = 1.0;                /* Vu[-2nu:-nu] = I */
    V[10 * NZ + 8] = -1.0; V[11 * NZ + 9] = -1.0;            /* Vu[-nu:] = -I */
}

static const double *track_s(const orc_problem *P, int tid) { return P->s_ref + (size_t)tid * P->nknots; }
static const double *track_k(const orc_problem *P, int tid) { return P->kappa_ref + (size_t)tid * P->nknots; }

void orc_build_qp(const orc_problem *P, const double *x, const double *u, const double *x0,
                  const double *yref, const double *yref_e, int tid, double *H, double *g,
                  double *A, double *Bm, double *b, double *dx0, double *R, double *dl, double *du)
{
    const int N = P->N;
    double V[NY * NZ];
    cost_selector(V);
    memset(H, 0, sizeof(double) * (N + 1) * NZ * NZ);
    memset(g, 0, sizeof(double) * (N + 1) * NZ);
    memset(R, 0, sizeof(double) * (N + 1) * NC * NZ);
    for (int k = 0; k <= N; k++) {
        const double *xk = x + k * NX;
        if (k < N) {
            const double *uk = u + k * NU;
            /* dynamics */
            double xn[NX];
            orc_rk4_sens(P->model, P->integrator, xk, uk, track_s(P, tid), track_k(P, tid), P->nknots, P->dt, P->M, xn,
                         A + k * NX * NX, Bm + k * NX * NU);
            for (int i = 0; i < NX; i++) b[k * NX + i] = xn[i] - x[(k + 1) * NX + i];
            /* cost: H = c V' W V, g = c V' W (V z - yref) */
            const double *W = P->W + (size_t)k * NY * NY;
            double zk[NZ], e[NY], We[NY], WV[NY * NZ];
            memcpy(zk, xk, sizeof(double) * NX); zk[8] = uk[0]; zk[9] = uk[1];
            for (int i = 0; i < NY; i++) {
                double acc = -yref[k * NY + i];
                for (int j = 0; j < NZ; j++) acc += V[i * NZ + j] * zk[j];
                e[i] = acc;
            }
            for (int i = 0; i < NY; i++) {
                double acc = 0;
                for (int j = 0; j < NY; j++) acc += W[i * NY + j] * e[j];
                We[i] = acc;
                for (int j = 0; j < NZ; j++) {
                    double a2 = 0;
                    for (int l = 0; l < NY; l++) a2 += W[i * NY + l] * V[l * NZ + j];
                    WV[i * NZ + j] = a2;
                }
            }
            for (int i = 0; i < NZ; i++) {
                double acc = 0;
                for (int l = 0; l < NY; l++) acc += V[l * NZ + i] * We[l];
                g[k * NZ + i] = P->cost_scale_stage * acc;
                for (int j = 0; j < NZ; j++) {
                    double a2 = 0;
                    for (int l = 0; l < NY; l++) a2 += V[l * NZ + i] * WV[l * NZ + j];
                    H[(k * NZ + i) * NZ + j] = P->cost_scale_stage * a2;
                }
            }
        } else {
            for (int i = 0; i < NX; i++) {
                double acc = 0;
                for (int j = 0; j < NX; j++) {
                    acc += P->W_e[i * NX + j] * (xk[j] - yref_e[j]);
                    H[(k * NZ + i) * NZ + j] = P->W_e[i * NX + j];
                }
                g[k * NZ + i] = acc;
            }
            /* keep the padded u-block non-singular; it never couples to anything */
            H[(k * NZ + 8) * NZ + 8] = 1.0; H[(k * NZ + 9) * NZ + 9] = 1.0;
        }
        /* constraints: rows 0..7 state bounds, 8..9 input bounds, 10..11 general rows, 12..13 track rows */
        for (int c = 0; c < NC; c++) { dl[k * NC + c] = -INFINITY; du[k * NC + c] = INFINITY; }
        if (k >= 1)
            for (int i = 0; i < NX; i++) {
                R[(k * NC + i) * NZ + i] = 1.0;
                double lb = P->lbx[k * NX + i], ub = P->ubx[k * NX + i];
                if (fabs(lb) < INF_BOUND) dl[k * NC + i] = lb - xk[i];
                if (fabs(ub) < INF_BOUND) du[k * NC + i] = ub - xk[i];
            }
        if (k < N) {
            const double *uk = u + k * NU;
            for (int i = 0; i < NU; i++) {
                R[(k * NC + 8 + i) * NZ + 8 + i] = 1.0;
                double lb = P->lbu[k * NU + i], ub = P->ubu[k * NU + i];
                if (fabs(lb) < INF_BOUND) dl[k * NC + 8 + i] = lb - uk[i];
                if (fabs(ub) < INF_BOUND) du[k * NC + 8 + i] = ub - uk[i];
            }
            for (int i = 0; i < ORC_NG; i++) {
                double val = 0;
                for (int j = 0; j < NX; j++) { double cj = P->C[(k * ORC_NG + i) * NX + j]; R[(k * NC + 10 + i) * NZ + j] = cj; val += cj * xk[j]; }
                for (int j = 0; j < NU; j++) { double dj = P->D[(k * ORC_NG + i) * NU + j]; R[(k * NC + 10 + i) * NZ + 8 + j] = dj; val += dj * uk[j]; }
                double lb = P->lg[k * ORC_NG + i], ub = P->ug[k * ORC_NG + i];
                if (fabs(lb) < INF_BOUND) dl[k * NC + 10 + i] = lb - val;
                if (fabs(ub) < INF_BOUND) du[k * NC + 10 + i] = ub - val;
            }
        }
        if (P->path_on && k >= 1) {
            /* h(x) and its gradient in (n, psi); d|psi|/dpsi = sign(psi) with sign(0) = 0 as CasADi differentiates fabs */
            const double n = xk[1], psi = xk[2], sg = (psi > 0.0) - (psi < 0.0);
            const double hl = 0.5 * P->car_L, hw = 0.5 * P->car_W;
            const double foot = -hl * sin(fabs(psi)), lat = hw * cos(psi);
            const double dfoot = -hl * cos(fabs(psi)) * sg, dlat = -hw * sin(psi);
            const double hval[ORC_NH] = {n + foot + lat - P->widths[tid * 2 + 0], -n - foot + lat - P->widths[tid * 2 + 1]};
            const double dn[ORC_NH] = {1.0, -1.0}, dpsi[ORC_NH] = {dfoot + dlat, -dfoot + dlat};
            for (int i = 0; i < ORC_NH; i++) {
                R[(k * NC + 12 + i) * NZ + 1] = dn[i];
                R[(k * NC + 12 + i) * NZ + 2] = dpsi[i];
                if (fabs(P->lh[i]) < INF_BOUND) dl[k * NC + 12 + i] = P->lh[i] - hval[i];
                if (fabs(P->uh[i]) < INF_BOUND) du[k * NC + 12 + i] = P->uh[i] - hval[i];
            }
        }
#if ORC_NC > 14
        /* row 14: the lateral-acceleration row of the kinematic model, stages 1..N-1 (no terminal row: con_h_expr_e has the track rows only,
         * old/generate_acaods_interface.py:209-212; x_0 is fixed) */
        if (P->alat_on && k >= 1 && k < N) {
            double grad[NX];
            const double aval = orc_alat(xk, grad);
            for (int j = 0; j < NX; j++) R[(k * NC + 14) * NZ + j] = grad[j];
            if (fabs(P->alat_lb) < INF_BOUND) dl[k * NC + 14] = P->alat_lb - aval;
            if (fabs(P->alat_ub) < INF_BOUND) du[k * NC + 14] = P->alat_ub - aval;
        }
#endif
    }
    for (int i = 0; i < NX; i++) dx0[i] = x0[i] - x[i];
}

/* sl_out (NS,28): slack values of the QP; g_out (NS,10): cost gradient at the linearisation point; dz_out (NS,10): QP step;
 * b_out (N,8): dynamics defects Phi(x_k,u_k) - x_{k+1} of the linearisation; pi0_out (8): multiplier of x_0 = x0, from the stage-0 stationarity row of the QP -- all optional (the globalised SQP needs them) */
static void rti_one(const orc_problem *P, double *x, double *u, const double *x0, const double *yref,
                    const double *yref_e, int tid, double *pi, double *lam, int *status, double *res, int *qp_iter,
                    double *sl_out, double *g_out, double *dz_out, double *pi0_out, double *b_out)
{
    const int N = P->N, NS = N + 1;
    double *H = malloc(sizeof(double) * NS * NZ * NZ), *g = malloc(sizeof(double) * NS * NZ);
    double *A = malloc(sizeof(double) * N * NX * NX), *Bm = malloc(sizeof(double) * N * NX * NU);
    double *b = malloc(sizeof(double) * N * NX), *R = malloc(sizeof(double) * NS * NC * NZ);
    double *dl = malloc(sizeof(double) * NS * NC), *du = malloc(sizeof(double) * NS * NC);
    double *dz = malloc(sizeof(double) * NS * NZ), *qpi = malloc(sizeof(double) * NS * NX);
    double *qlam = malloc(sizeof(double) * NS * 2 * NC), *qt = malloc(sizeof(double) * NS * 2 * NC);
    double dx0[NX], stats[8];

    orc_build_qp(P, x, u, x0, yref, yref_e, tid, H, g, A, Bm, b, dx0, R, dl, du);

    /* NLP KKT residuals at the current iterate with the incoming multipliers */
    double r_stat = 0, r_eq = 0, r_ineq = 0, r_comp = 0;
    for (int i = 0; i < NX; i++) r_eq = fmax(r_eq, fabs(dx0[i]));
    for (int k = 0; k <= N; k++) {
        for (int j = 0; j < ((k < N) ? NZ : NX); j++) {
            if (k == 0 && j < NX) continue;
            double acc = g[k * NZ + j];
            if (k < N)
                for (int l = 0; l < NX; l++) {
                    double ab = (j < NX) ? A[(k * NX + l) * NX + j] : Bm[(k * NX + l) * NU + (j - NX)];
                    acc += ab * pi[(k + 1) * NX + l];
                }
            if (j < NX) acc -= pi[k * NX + j];
            for (int c = 0; c < NC; c++)
                acc -= R[(k * NC + c) * NZ + j] * (lam[k * 2 * NC + c] - lam[k * 2 * NC + NC + c]);
            r_stat = fmax(r_stat, fabs(acc));
        }
        if (k < N) for (int i = 0; i < NX; i++) r_eq = fmax(r_eq, fabs(b[k * NX + i]));
        for (int c = 0; c < NC; c++) {
            double ll = dl[k * NC + c], uu = du[k * NC + c]; /* = bound - c(z): slack is -ll, uu */
            /* soft sides may be violated: they do not count as infeasibility */
            const int sl_ = P->soft_Z && P->soft_Z[k * 2 * NC + c] >= 0.0, su_ = P->soft_Z && P->soft_Z[k * 2 * NC + NC + c] >= 0.0;
            if (fabs(ll) < INF_BOUND && !sl_) { r_ineq = fmax(r_ineq, ll); r_comp = fmax(r_comp, fabs(lam[k * 2 * NC + c] * ll)); }
            if (fabs(uu) < INF_BOUND && !su_) { r_ineq = fmax(r_ineq, -uu); r_comp = fmax(r_comp, fabs(lam[k * 2 * NC + NC + c] * uu)); }
        }
    }
    res[0] = r_stat; res[1] = r_eq; res[2] = r_ineq; res[3] = r_comp;

    int iters = 0;
    int qs = orc_qp_solve_soft(N, H, g, A, Bm, b, dx0, R, dl, du, P->soft_z, P->soft_Z, P->ipm_iter_max, P->ipm_tol,
                               P->ipm_mu0, P->ipm_tau0, dz, qpi, qlam, qt, sl_out, stats, &iters);
    *qp_iter = iters;
    int st = 0;
    if (qs == 3) st = 1;                 /* NaN */
    else if (qs == 2 || qs == 4) st = 4; /* QP failure: min step, or max iter far from converged */
    /* qs == 1 (max iter, loosely converged) is tolerated in RTI, as acados does */
    int bad = 0;
    for (int i = 0; i < NS * NZ; i++) if (!isfinite(dz[i])) bad = 1;
    if (bad) st = 1;
    if (g_out) memcpy(g_out, g, sizeof(double) * NS * NZ);
    if (b_out) memcpy(b_out, b, sizeof(double) * N * NX);
    if (dz_out) memcpy(dz_out, dz, sizeof(double) * NS * NZ);
    if (pi0_out)    /* H_0 dz_0 + g_0 + A_0' pi_1 - pi_0 - R_0' (lam_l - lam_u) = 0, state rows */
        for (int j = 0; j < NX; j++) {
            double acc = g[j];
            for (int l = 0; l < NZ; l++) acc += H[j * NZ + l] * dz[l];
            for (int l = 0; l < NX; l++) acc += A[l * NX + j] * qpi[NX + l];
            for (int c = 0; c < NC; c++) acc -= R[c * NZ + j] * (qlam[c] - qlam[NC + c]);
            pi0_out[j] = acc;
        }
    if (st == 0) { /* a failed instance keeps its iterate: it must not poison later steps */
        for (int k = 0; k <= N; k++) {
            for (int i = 0; i < NX; i++) x[k * NX + i] += dz[k * NZ + i];
            if (k < N) for (int i = 0; i < NU; i++) u[k * NU + i] += dz[k * NZ + 8 + i];
        }
        memcpy(pi, qpi, sizeof(double) * NS * NX);
        for (int i = 0; i < NX; i++) pi[i] = 0.0; /* x_0 is eliminated: pi_0 is not defined */
        memcpy(lam, qlam, sizeof(double) * NS * 2 * NC);
    }
    *status = st;
    free(H); free(g); free(A); free(Bm); free(b); free(R); free(dl); free(du); free(dz); free(qpi); free(qlam); free(qt);
}

void orc_rti_step(const orc_problem *P, int B, double *x, double *u, const double *x0,
                  const double *yref, const double *yref_e, const int *track_id, double *pi,
                  double *lam, int *status, double *res, int *qp_iter, int nthreads)
{
    const int N = P->N;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
    for (int i = 0; i < B; i++) {
        rti_one(P, x + (size_t)i * (N + 1) * NX, u + (size_t)i * N * NU, x0 + (size_t)i * NX,
                yref + (size_t)i * N * NY, yref_e + (size_t)i * NX, track_id ? track_id[i] : 0,
                pi + (size_t)i * (N + 1) * NX, lam + (size_t)i * (N + 1) * 2 * NC, status + i,
                res + (size_t)i * 4, qp_iter + i, 0, 0, 0, 0, 0);
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Globalised SQP: python/main.py:230-237 asks acados for nlp_solver_type "SQP", nlp_solver_max_iter 2 and
 * globalization "MERIT_BACKTRACKING".  acados (third party, unpinned, absent) documents that option as a backtracking line
 * search on the l1 merit function
 *     m(w) = cost(w) + sum_i w_pi_i |dynamics defect_i| + sum_j w_lam_j max(0, violation_j)
 * whose weights follow the QP multipliers: |mult| on the first iteration, then max(|mult|, (w + |mult|) / 2); trial steps
 * alpha = 1, rho, rho^2, ... (rho = alpha_reduction 0.7) down to alpha_min 0.05, accepted on plain decrease
 * (line_search_use_sufficient_descent = 0, the default) or on the Armijo condition m(alpha) - m(0) <= eps alpha D with D the
 * directional derivative of the merit along the step; primal AND dual variables take the step alpha (full_step_dual = 0).
 * Each iteration first tests the four KKT residuals at the iterate against the tolerances (status 0), a QP failure ends the
 * solve with status 4, running out of iterations with status 2 (ACADOS_MAXITER; the reference accepts 0 and 2,
 * python/main.py:326).  PARITY UNPINNED like the rest of this oracle.
 * ------------------------------------------------------------------------------------------------------------------ */

/* bound - c(w) of every constraint row (-+inf = absent), as orc_build_qp linearises them */
static void constraint_gaps(const orc_problem *P, const double *x, const double *u, int tid, double *dl, double *du)
{
    const int N = P->N;
    for (int k = 0; k <= N; k++) {
        const double *xk = x + k * NX;
        for (int c = 0; c < NC; c++) { dl[k * NC + c] = -INFINITY; du[k * NC + c] = INFINITY; }
        double cv[NC];
        for (int i = 0; i < NX; i++) cv[i] = xk[i];
        if (k < N) {
            const double *uk = u + k * NU;
            cv[8] = uk[0]; cv[9] = uk[1];
            for (int i = 0; i < ORC_NG; i++) {
                double val = 0;
                for (int j = 0; j < NX; j++) val += P->C[(k * ORC_NG + i) * NX + j] * xk[j];
                for (int j = 0; j < NU; j++) val += P->D[(k * ORC_NG + i) * NU + j] * uk[j];
                cv[10 + i] = val;
            }
        }
        if (P->path_on && k >= 1) {
            const double n = xk[1], psi = xk[2];
            const double foot = -0.5 * P->car_L * sin(fabs(psi)), lat = 0.5 * P->car_W * cos(psi);
            cv[12] = n + foot + lat - P->widths[tid * 2 + 0];
            cv[13] = -n - foot + lat - P->widths[tid * 2 + 1];
        }
#if ORC_NC > 14
        cv[14] = (P->alat_on && k >= 1 && k < N) ? orc_alat(xk, NULL) : 0.0;
#endif
        for (int c = 0; c < NC; c++) {
            double lb = INFINITY, ub = INFINITY;     /* |bound| >= 1e20: absent */
            if (c < 8) { if (k >= 1) { lb = P->lbx[k * NX + c]; ub = P->ubx[k * NX + c]; } }
            else if (c < 10) { if (k < N) { lb = P->lbu[k * NU + c - 8]; ub = P->ubu[k * NU + c - 8]; } }
            else if (c < 12) { if (k < N) { lb = P->lg[k * ORC_NG + c - 10]; ub = P->ug[k * ORC_NG + c - 10]; } }
            else if (c < 14) { if (P->path_on && k >= 1) { lb = P->lh[c - 12]; ub = P->uh[c - 12]; } }
            else if (P->alat_on && k >= 1 && k < N) { lb = P->alat_lb; ub = P->alat_ub; }
            if (fabs(lb) < INF_BOUND) dl[k * NC + c] = lb - cv[c];
            if (fabs(ub) < INF_BOUND) du[k * NC + c] = ub - cv[c];
        }
    }
}

/* l1 merit at (x, u, sl) with weights wpi (NS,8) [row 0: the constraint x_0 = x0] and wlam (NS,28); sl may be NULL (all hard);
 * defect (N,8): the dynamics defects if they are already known (the linearisation point), else NULL = integrate */
static double merit_eval(const orc_problem *P, const double *x, const double *u, const double *sl, const double *x0,
                         const double *yref, const double *yref_e, int tid, const double *wpi, const double *wlam,
                         double *dl, double *du, const double *defect)
{
    const int N = P->N;
    double V[NY * NZ];
    cost_selector(V);
    double cost = 0.0, eq = 0.0, ineq = 0.0;
    for (int i = 0; i < NX; i++) eq += wpi[i] * fabs(x0[i] - x[i]);
    for (int k = 0; k <= N; k++) {
        const double *xk = x + k * NX;
        if (k < N) {
            const double *uk = u + k * NU, *W = P->W + (size_t)k * NY * NY;
            double zk[NZ], e[NY], xn[NX];
            memcpy(zk, xk, sizeof(double) * NX); zk[8] = uk[0]; zk[9] = uk[1];
            for (int i = 0; i < NY; i++) {
                double acc = -yref[k * NY + i];
                for (int j = 0; j < NZ; j++) acc += V[i * NZ + j] * zk[j];
                e[i] = acc;
            }
            double q = 0.0;
            for (int i = 0; i < NY; i++) {
                double acc = 0;
                for (int j = 0; j < NY; j++) acc += W[i * NY + j] * e[j];
                q += e[i] * acc;
            }
            cost += 0.5 * P->cost_scale_stage * q;
            if (defect)
                for (int i = 0; i < NX; i++) eq += wpi[(k + 1) * NX + i] * fabs(defect[k * NX + i]);
            else {
                orc_rk4(P->model, P->integrator, xk, uk, track_s(P, tid), track_k(P, tid), P->nknots, P->dt, P->M, xn);
                for (int i = 0; i < NX; i++) eq += wpi[(k + 1) * NX + i] * fabs(xn[i] - x[(k + 1) * NX + i]);
            }
        } else {
            double q = 0.0;
            for (int i = 0; i < NX; i++) {
                double acc = 0;
                for (int j = 0; j < NX; j++) acc += P->W_e[i * NX + j] * (xk[j] - yref_e[j]);
                q += (xk[i] - yref_e[i]) * acc;
            }
            cost += 0.5 * q;
        }
    }
    constraint_gaps(P, x, u, tid, dl, du);
    for (int k = 0; k <= N; k++)
        for (int c = 0; c < 2 * NC; c++) {
            const int up = c >= NC, r = up ? c - NC : c;
            const double gap = up ? du[k * NC + r] : dl[k * NC + r];        /* bound - c(w) */
            if (!(fabs(gap) < INFINITY)) continue;
            double viol = up ? -gap : gap;                                   /* > 0: violated */
            if (P->soft_Z && P->soft_Z[k * 2 * NC + c] >= 0.0) {
                const double sv = sl ? sl[k * 2 * NC + c] : 0.0;
                cost += P->soft_z[k * 2 * NC + c] * sv + 0.5 * P->soft_Z[k * 2 * NC + c] * sv * sv;
                viol -= sv;
            }
            ineq += wlam[k * 2 * NC + c] * fmax(0.0, viol);
        }
    return cost + eq + ineq;
}

static void sqp_one(const orc_problem *P, const orc_sqp_opts *O, double *x, double *u, const double *x0, const double *yref,
                    const double *yref_e, int tid, double *pi, double *lam, double *sl, int *status, double *res, int *qp_iter,
                    int *sqp_iter, double *alpha_out)
{
    const int N = P->N, NS = N + 1, nx = NS * NX, nu = N * NU, nl = NS * 2 * NC;
    double *xp = malloc(sizeof(double) * nx), *up = malloc(sizeof(double) * nu), *pip = malloc(sizeof(double) * nx);
    double *lamp = malloc(sizeof(double) * nl), *slp = malloc(sizeof(double) * nl), *sln = malloc(sizeof(double) * nl);
    double *xt = malloc(sizeof(double) * nx), *ut = malloc(sizeof(double) * nu), *slt = malloc(sizeof(double) * nl);
    double *wpi = calloc(nx, sizeof(double)), *wlam = calloc(nl, sizeof(double));
    double *g = malloc(sizeof(double) * NS * NZ), *dz = malloc(sizeof(double) * NS * NZ);
    double *dl = malloc(sizeof(double) * NS * NC), *du = malloc(sizeof(double) * NS * NC), *b0 = malloc(sizeof(double) * N * NX);
    double pi0[NX];
    int st = 2, iters_total = 0, it;
    double alpha = 1.0;
    for (it = 0; it < O->max_iter; it++) {
        memcpy(xp, x, sizeof(double) * nx); memcpy(up, u, sizeof(double) * nu);
        memcpy(pip, pi, sizeof(double) * nx); memcpy(lamp, lam, sizeof(double) * nl); memcpy(slp, sl, sizeof(double) * nl);
        int qs, qi;
        rti_one(P, x, u, x0, yref, yref_e, tid, pi, lam, &qs, res, &qi, sln, g, dz, pi0, b0);
        iters_total += qi;
        if (res[0] <= O->tol[0] && res[1] <= O->tol[1] && res[2] <= O->tol[2] && res[3] <= O->tol[3]) {
            /* converged at the linearisation point: the iterate stays */
            memcpy(x, xp, sizeof(double) * nx); memcpy(u, up, sizeof(double) * nu);
            memcpy(pi, pip, sizeof(double) * nx); memcpy(lam, lamp, sizeof(double) * nl);
            st = 0; break;
        }
        if (qs != 0) { st = qs; break; }       /* rti_one left the iterate untouched */
        if (O->globalization == 0) { memcpy(sl, sln, sizeof(double) * nl); alpha = 1.0; continue; }
        /* merit weights from the QP multipliers (pi_0: the multiplier of x_0 = x0) */
        for (int i = 0; i < nx; i++) {
            const double m = fabs(i < NX ? pi0[i] : pi[i]);
            wpi[i] = (it == 0) ? m : fmax(m, 0.5 * (wpi[i] + m));
        }
        for (int i = 0; i < nl; i++) {
            const double m = fabs(lam[i]);
            wlam[i] = (it == 0) ? m : fmax(m, 0.5 * (wlam[i] + m));
        }
        const double m0 = merit_eval(P, xp, up, slp, x0, yref, yref_e, tid, wpi, wlam, dl, du, b0);   /* defects: from the linearisation */
        /* directional derivative of the merit along the step: grad cost . d - (weighted infeasibility at alpha = 0),
         * the linearised constraints hold at the full step */
        double D = 0.0;
        if (O->use_sufficient_descent) {
            double eqi = m0;
            for (int k = 0; k <= N; k++)
                for (int j = 0; j < ((k < N) ? NZ : NX); j++) D += g[k * NZ + j] * dz[k * NZ + j];
            /* infeasibility part of m0 = m0 - cost(0): recompute the cost with zero weights */
            double *zw = calloc(nx > nl ? nx : nl, sizeof(double));
            const double c0 = merit_eval(P, xp, up, slp, x0, yref, yref_e, tid, zw, zw, dl, du, b0);
            free(zw);
            eqi -= c0;
            if (P->soft_Z)
                for (int i = 0; i < nl; i++)
                    if (P->soft_Z[i] >= 0.0) D += (P->soft_z[i] + P->soft_Z[i] * slp[i]) * (sln[i] - slp[i]);
            D -= eqi;
            if (D > 0.0) D = 0.0;
        }
        alpha = 1.0;
        for (;;) {
            for (int i = 0; i < nx; i++) xt[i] = xp[i] + alpha * (x[i] - xp[i]);
            for (int i = 0; i < nu; i++) ut[i] = up[i] + alpha * (u[i] - up[i]);
            for (int i = 0; i < nl; i++) slt[i] = slp[i] + alpha * (sln[i] - slp[i]);
            const double m1 = merit_eval(P, xt, ut, slt, x0, yref, yref_e, tid, wpi, wlam, dl, du, 0);
            if (O->use_sufficient_descent ? (m1 - m0 <= O->eps_sufficient_descent * alpha * D) : (m1 < m0)) break;
            alpha *= O->alpha_reduction;
            if (alpha < O->alpha_min) { alpha = O->alpha_min; break; }
        }
        for (int i = 0; i < nx; i++) x[i] = xp[i] + alpha * (x[i] - xp[i]);
        for (int i = 0; i < nu; i++) u[i] = up[i] + alpha * (u[i] - up[i]);
        for (int i = 0; i < nl; i++) sl[i] = slp[i] + alpha * (sln[i] - slp[i]);
        if (!O->full_step_dual) {
            for (int i = 0; i < nx; i++) pi[i] = pip[i] + alpha * (pi[i] - pip[i]);
            for (int i = 0; i < nl; i++) lam[i] = lamp[i] + alpha * (lam[i] - lamp[i]);
        }
    }
    *status = st; *qp_iter = iters_total; *sqp_iter = (it < O->max_iter) ? it + (st != 0) : it; *alpha_out = alpha;
    free(xp); free(up); free(pip); free(lamp); free(slp); free(sln); free(xt); free(ut); free(slt); free(wpi); free(wlam);
    free(g); free(dz); free(dl); free(du); free(b0);
}

void orc_sqp_solve(const orc_problem *P, const orc_sqp_opts *O, int B, double *x, double *u, const double *x0,
                   const double *yref, const double *yref_e, const int *track_id, double *pi, double *lam, double *sl,
                   int *status, double *res, int *qp_iter, int *sqp_iter, double *alpha, int nthreads)
{
    const int N = P->N;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
    for (int i = 0; i < B; i++)
        sqp_one(P, O, x + (size_t)i * (N + 1) * NX, u + (size_t)i * N * NU, x0 + (size_t)i * NX, yref + (size_t)i * N * NY,
                yref_e + (size_t)i * NX, track_id ? track_id[i] : 0, pi + (size_t)i * (N + 1) * NX,
                lam + (size_t)i * (N + 1) * 2 * NC, sl + (size_t)i * (N + 1) * 2 * NC, status + i, res + (size_t)i * 4,
                qp_iter + i, sqp_iter + i, alpha + i);
}

void orc_linearize(const orc_problem *P, int B, const double *x, const double *u,
                   const int *track_id, double *A, double *Bm, double *b, int nthreads)
{
    const int N = P->N;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
    for (int i = 0; i < B; i++) {
        int tid = track_id ? track_id[i] : 0;
        for (int k = 0; k < N; k++) {
            const double *xk = x + ((size_t)i * (N + 1) + k) * NX, *uk = u + ((size_t)i * N + k) * NU;
            double xn[NX];
            orc_rk4_sens(P->model, P->integrator, xk, uk, track_s(P, tid), track_k(P, tid), P->nknots, P->dt, P->M, xn,
                         A + ((size_t)i * N + k) * NX * NX, Bm + ((size_t)i * N + k) * NX * NU);
            for (int j = 0; j < NX; j++) b[((size_t)i * N + k) * NX + j] = xn[j] - xk[NX + j];
        }
    }
}

/* python/main.py:297-322: yref ramp from s0 = x0[0]; shift of the previous prediction */
void orc_prepare_step(int N, int B, const double *x0, double s_target, double *x, double *u,
                      double *yref, double *yref_e)
{
    for (int i = 0; i < B; i++) {
        double *xi = x + (size_t)i * (N + 1) * NX, *ui = u + (size_t)i * N * NU;
        double *yi = yref + (size_t)i * N * NY, *ye = yref_e + (size_t)i * NX;
        double s0 = x0[(size_t)i * NX];
        memset(yi, 0, sizeof(double) * N * NY);
        memset(ye, 0, sizeof(double) * NX);
        for (int j = 0; j < N; j++) yi[j * NY] = s0 + s_target * j / N;
        ye[0] = s0 + s_target;
        /* x_j <- x_pred[j+1], u_j <- u_pred[j+1] (j < N-1); x_{N-1}, x_N <- x_pred[N]; u_{N-1} <- 0 */
        for (int j = 0; j < N - 1; j++) {
            memcpy(xi + j * NX, xi + (j + 1) * NX, sizeof(double) * NX);
            memcpy(ui + j * NU, ui + (j + 1) * NU, sizeof(double) * NU);
        }
        memcpy(xi + (N - 1) * NX, xi + N * NX, sizeof(double) * NX);
        ui[(N - 1) * NU] = 0.0; ui[(N - 1) * NU + 1] = 0.0;
    }
}

void orc_sim_step(const orc_problem *P, int B, int model, int M, const double *x, const double *u,
                  const int *track_id, double *xnext, int nthreads)
{
    orc_sim_step_integ(P, B, model, ORC_INTEG_RK4, M, x, u, track_id, xnext, nthreads);
}

void orc_sim_step_integ(const orc_problem *P, int B, int model, int integrator, int M, const double *x, const double *u,
                        const int *track_id, double *xnext, int nthreads)
{
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int i = 0; i < B; i++) {
        int tid = track_id ? track_id[i] : 0;
        orc_rk4(model, integrator, x + (size_t)i * NX, u + (size_t)i * NU, track_s(P, tid), track_k(P, tid),
                P->nknots, P->dt, M, xnext + (size_t)i * NX);
    }
}
