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
    }
    for (int i = 0; i < NX; i++) dx0[i] = x0[i] - x[i];
}

static void rti_one(const orc_problem *P, double *x, double *u, const double *x0, const double *yref,
                    const double *yref_e, int tid, double *pi, double *lam, int *status, double *res, int *qp_iter)
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
                               P->ipm_mu0, P->ipm_tau0, dz, qpi, qlam, qt, 0, stats, &iters);
    *qp_iter = iters;
    int st = 0;
    if (qs == 3) st = 1;                 /* NaN */
    else if (qs == 2 || qs == 4) st = 4; /* QP failure: min step, or max iter far from converged */
    /* qs == 1 (max iter, loosely converged) is tolerated in RTI, as acados does */
    int bad = 0;
    for (int i = 0; i < NS * NZ; i++) if (!isfinite(dz[i])) bad = 1;
    if (bad) st = 1;
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
                res + (size_t)i * 4, qp_iter + i);
    }
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
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int i = 0; i < B; i++) {
        int tid = track_id ? track_id[i] : 0;
        orc_rk4(model, ORC_INTEG_RK4, x + (size_t)i * NX, u + (size_t)i * NU, track_s(P, tid), track_k(P, tid),
                P->nknots, P->dt, M, xnext + (size_t)i * NX);
    }
}
