/*
 * ihm2_oracle_qp.c -- CPU ORACLE (test infrastructure, NOT the product).
 * Stage-wise convex QP by a Riccati-based primal-dual interior-point method
 * (Mehrotra predictor-corrector), dense and readable.
 *
 * The reference delegates this to HPIPM through acados (python/main.py:228-233:
 * PARTIAL_CONDENSING_HPIPM with the default horizon = no condensing, hpipm_mode SPEED_ABS);
 * HPIPM is third party, unpinned and absent. The QP has a unique solution
 * (H_k > 0), so parity is on the solution and its KKT residual, not on HPIPM's iterate path
 * (SURVEY.md Appendix D).
 *
 * QP (k = 0..N; z_k = (dx_k, du_k) in R^10, z_N = dx_N in R^8, padded to 10):
 *   min  sum_k 1/2 z_k' H_k z_k + g_k' z_k
 *   s.t. dx_0 = dx0,  dx_{k+1} = A_k dx_k + B_k du_k + b_k,
 *        dl_{k,c} <= R_{k,c}' z_k <= du_{k,c}       c = 0..11  (+-inf = absent)
 * Lagrangian sign convention: + pi_{k+1}'(A dx + B du + b - dx_{k+1}) - lam_l'(Rz - dl) - lam_u'(du - Rz).
 */
#include "ihm2_oracle.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

int orc_debug = 0; /* tests may set this to trace IPM iterations on stderr */

#define NX ORC_NX
#define NU ORC_NU
#define NZ ORC_NZ
#define NC ORC_NC
#define INF_BOUND 1e20

static int is_fin(double v) { return fabs(v) < INF_BOUND; }

typedef struct {
    int N;
    const double *H, *g, *A, *Bm, *b, *R, *dl, *du;
    /* factorisation, per stage */
    double P[ORC_NMAX + 1][NX * NX];
    double Gux[ORC_NMAX][NU * NX];
    double Guu_inv[ORC_NMAX][NU * NU];
    double Ht[ORC_NMAX + 1][NZ * NZ]; /* barrier-augmented Hessian */
} qp_work;

/* backward Riccati sweep. do_factor: recompute P, Gux, Guu_inv from Ht. Always: vector part
 * from gt (modified gradient) and rb (dynamics residual). Outputs p[k] (N+1,8), kff[k] (N,2). */
static void riccati_backward(qp_work *w, int do_factor, const double *gt, const double *rb,
                             double *p, double *kff)
{
    const int N = w->N;
    if (do_factor)
        for (int i = 0; i < NX; i++)
            for (int j = 0; j < NX; j++) w->P[N][i * NX + j] = w->Ht[N][i * NZ + j];
    for (int i = 0; i < NX; i++) p[N * NX + i] = gt[N * NZ + i];
    for (int k = N - 1; k >= 0; k--) {
        const double *A = w->A + k * NX * NX, *Bm = w->Bm + k * NX * NU;
        const double *Pn = w->P[k + 1];
        double AB[NX * NZ];
        for (int i = 0; i < NX; i++) {
            for (int j = 0; j < NX; j++) AB[i * NZ + j] = A[i * NX + j];
            for (int j = 0; j < NU; j++) AB[i * NZ + NX + j] = Bm[i * NU + j];
        }
        if (do_factor) {
            double PAB[NX * NZ], G[NZ * NZ];
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NZ; j++) {
                    double acc = 0;
                    for (int l = 0; l < NX; l++) acc += Pn[i * NX + l] * AB[l * NZ + j];
                    PAB[i * NZ + j] = acc;
                }
            for (int i = 0; i < NZ; i++)
                for (int j = 0; j < NZ; j++) {
                    double acc = w->Ht[k][i * NZ + j];
                    for (int l = 0; l < NX; l++) acc += AB[l * NZ + i] * PAB[l * NZ + j];
                    G[i * NZ + j] = acc;
                }
            /* Guu (2x2) inverse */
            double a = G[8 * NZ + 8], bq = G[8 * NZ + 9], c = G[9 * NZ + 8], d = G[9 * NZ + 9];
            double det = a * d - bq * c;
            double *Gi = w->Guu_inv[k];
            Gi[0] = d / det; Gi[1] = -bq / det; Gi[2] = -c / det; Gi[3] = a / det;
            for (int i = 0; i < NU; i++)
                for (int j = 0; j < NX; j++) w->Gux[k][i * NX + j] = G[(NX + i) * NZ + j];
            /* P_k = Gxx - Gux' Guu^-1 Gux */
            double K[NU * NX];
            for (int i = 0; i < NU; i++)
                for (int j = 0; j < NX; j++)
                    K[i * NX + j] = Gi[i * NU + 0] * w->Gux[k][0 * NX + j] + Gi[i * NU + 1] * w->Gux[k][1 * NX + j];
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NX; j++)
                    w->P[k][i * NX + j] = G[i * NZ + j] - (w->Gux[k][0 * NX + i] * K[0 * NX + j] + w->Gux[k][1 * NX + i] * K[1 * NX + j]);
            /* symmetrise */
            for (int i = 0; i < NX; i++)
                for (int j = i + 1; j < NX; j++) {
                    double s = 0.5 * (w->P[k][i * NX + j] + w->P[k][j * NX + i]);
                    w->P[k][i * NX + j] = s; w->P[k][j * NX + i] = s;
                }
        }
        /* vector part */
        double h[NX], gz[NZ];
        for (int i = 0; i < NX; i++) {
            double acc = p[(k + 1) * NX + i];
            for (int l = 0; l < NX; l++) acc += Pn[i * NX + l] * rb[k * NX + l];
            h[i] = acc;
        }
        for (int j = 0; j < NZ; j++) {
            double acc = gt[k * NZ + j];
            for (int l = 0; l < NX; l++) acc += AB[l * NZ + j] * h[l];
            gz[j] = acc;
        }
        const double *Gi = w->Guu_inv[k];
        kff[k * NU + 0] = Gi[0] * gz[8] + Gi[1] * gz[9];
        kff[k * NU + 1] = Gi[2] * gz[8] + Gi[3] * gz[9];
        for (int i = 0; i < NX; i++)
            p[k * NX + i] = gz[i] - (w->Gux[k][0 * NX + i] * kff[k * NU + 0] + w->Gux[k][1 * NX + i] * kff[k * NU + 1]);
    }
}

/* forward sweep: dz (N+1,10), dpi (N+1,8); delta of x_0 is zero (x_0 stays at dx0). */
static void riccati_forward(const qp_work *w, const double *rb, const double *p, const double *kff,
                            double *dz, double *dpi)
{
    const int N = w->N;
    double dx[NX];
    memset(dx, 0, sizeof dx);
    for (int k = 0; k < N; k++) {
        const double *A = w->A + k * NX * NX, *Bm = w->Bm + k * NX * NU;
        const double *Gi = w->Guu_inv[k];
        double t0 = 0, t1 = 0, du[NU];
        for (int j = 0; j < NX; j++) { t0 += w->Gux[k][j] * dx[j]; t1 += w->Gux[k][NX + j] * dx[j]; }
        du[0] = -(Gi[0] * t0 + Gi[1] * t1) - kff[k * NU + 0];
        du[1] = -(Gi[2] * t0 + Gi[3] * t1) - kff[k * NU + 1];
        for (int i = 0; i < NX; i++) dz[k * NZ + i] = dx[i];
        dz[k * NZ + 8] = du[0]; dz[k * NZ + 9] = du[1];
        for (int i = 0; i < NX; i++) {
            double acc = p[k * NX + i];
            for (int l = 0; l < NX; l++) acc += w->P[k][i * NX + l] * dx[l];
            dpi[k * NX + i] = acc;
        }
        double dxn[NX];
        for (int i = 0; i < NX; i++) {
            double acc = rb[k * NX + i];
            for (int l = 0; l < NX; l++) acc += A[i * NX + l] * dx[l];
            acc += Bm[i * NU + 0] * du[0] + Bm[i * NU + 1] * du[1];
            dxn[i] = acc;
        }
        memcpy(dx, dxn, sizeof dx);
    }
    for (int i = 0; i < NX; i++) dz[N * NZ + i] = dx[i];
    dz[N * NZ + 8] = 0; dz[N * NZ + 9] = 0;
    for (int i = 0; i < NX; i++) {
        double acc = p[N * NX + i];
        for (int l = 0; l < NX; l++) acc += w->P[N][i * NX + l] * dx[l];
        dpi[N * NX + i] = acc;
    }
}

/* One-sided view of the constraints: index i = k*24 + c (c < 12: lower side, sigma = +1, bound dl) or
 * k*24 + 12 + c (upper side, sigma = -1, bound du).  t_i = sigma (R z) [+ s_i] - sigma d_i >= 0.
 * SOFT sides (soft_Z[i] >= 0; reference: old/generate_acaods_interface.py:380-395, zl/zu/Zl/Zu) carry a slack
 * variable s_i >= 0 with cost soft_z[i] s_i + 1/2 soft_Z[i] s_i^2 and its own multiplier lam_s_i; the slack
 * block is eliminated from the Newton system in closed form:
 *   D = Z + lam/t + lam_s/s,  gamma_eff = (lam/t)(Z + lam_s/s)/D,  coef_eff = c1 - (lam/t)(rs + c1 + c2)/D,
 *   ds = -(rs + c1 + c2)/D - (lam/t)/D * sigma R dz,  c1 = (rm1 + lam rd)/t, c2 = rm2/s, rs = Z s + z - lam - lam_s. */
int orc_qp_solve_soft(int N, const double *H, const double *g, const double *A, const double *Bm,
                      const double *b, const double *dx0, const double *R, const double *dl,
                      const double *du, const double *soft_z, const double *soft_Z, int iter_max, double tol,
                      double mu0, double tau0, double *z, double *pi, double *lam, double *t, double *sl,
                      double *stats, int *iters)
{
    static _Thread_local qp_work w;
    w.N = N; w.H = H; w.g = g; w.A = A; w.Bm = Bm; w.b = b; w.R = R; w.dl = dl; w.du = du;
    const int NS = N + 1, NI = NS * 2 * NC;
    double rg[(ORC_NMAX + 1) * NZ], rb[ORC_NMAX * NX], gt[(ORC_NMAX + 1) * NZ];
    double p[(ORC_NMAX + 1) * NX], kff[ORC_NMAX * NU];
    double dz[(ORC_NMAX + 1) * NZ], dpi[(ORC_NMAX + 1) * NX];
    double Rz[(ORC_NMAX + 1) * NC];
    double rd[(ORC_NMAX + 1) * 2 * NC], rm[(ORC_NMAX + 1) * 2 * NC], bnd[(ORC_NMAX + 1) * 2 * NC];
    double dlam[(ORC_NMAX + 1) * 2 * NC], dt[(ORC_NMAX + 1) * 2 * NC];
    double dlam_a[(ORC_NMAX + 1) * 2 * NC], dt_a[(ORC_NMAX + 1) * 2 * NC];
    /* slack block of the soft sides */
    double s[(ORC_NMAX + 1) * 2 * NC], lams[(ORC_NMAX + 1) * 2 * NC], rs[(ORC_NMAX + 1) * 2 * NC], rm2[(ORC_NMAX + 1) * 2 * NC];
    double ds[(ORC_NMAX + 1) * 2 * NC], dlams[(ORC_NMAX + 1) * 2 * NC], ds_a[(ORC_NMAX + 1) * 2 * NC], dlams_a[(ORC_NMAX + 1) * 2 * NC];
    unsigned char act[(ORC_NMAX + 1) * 2 * NC], soft[(ORC_NMAX + 1) * 2 * NC];
    int status = 1, it = 0, m_act = 0, exact_mode = 0;
#define SGN(i) ((((i) % (2 * NC)) < NC) ? 1.0 : -1.0)
#define ROW(i) (((i) / (2 * NC)) * NC + ((i) % NC))

    /* initial point: z = 0 except x_0 = dx0; pi = 0; slacks clamped from below */
    memset(z, 0, sizeof(double) * NS * NZ);
    memset(pi, 0, sizeof(double) * NS * NX);
    for (int i = 0; i < NX; i++) z[i] = dx0[i];
    /* problem scales: tolerances and the initial barrier parameter are relative to them */
    double sg = 1.0, sb = 1.0;
    for (int k = 0; k < NS; k++)
        for (int j = 0; j < ((k < N) ? NZ : NX); j++) sg = fmax(sg, fabs(g[k * NZ + j]));
    for (int i = 0; i < N * NX; i++) sb = fmax(sb, fabs(b[i]));
    for (int i = 0; i < NX; i++) sb = fmax(sb, fabs(dx0[i]));
    const double tol_g = tol * sg, tol_b = tol * sb, tol_d = tol * sb, tol_m = tol * sg;
    const double mu_floor = 0.1 * tol_m;
    mu0 *= sg;
    for (int k = 0; k < NS; k++)
        for (int c = 0; c < NC; c++) {
            double rz = 0;
            for (int j = 0; j < NZ; j++) rz += R[(k * NC + c) * NZ + j] * z[k * NZ + j];
            Rz[k * NC + c] = rz;
        }
    for (int i = 0; i < NI; i++) {
        const int k = i / (2 * NC), c = i % NC;
        bnd[i] = (SGN(i) > 0) ? dl[k * NC + c] : du[k * NC + c];
        act[i] = (unsigned char)is_fin(bnd[i]);
        soft[i] = (unsigned char)(act[i] && soft_Z && soft_Z[i] >= 0.0);
        s[i] = 0.0; lams[i] = 0.0; t[i] = 1.0; lam[i] = 0.0;
        if (sl) sl[i] = 0.0;
    }
    for (int i = 0; i < NI; i++) {
        if (!act[i]) continue;
        const int k = i / (2 * NC), c = i % NC;
        const double slack = SGN(i) * (Rz[ROW(i)] - bnd[i]);
        if (soft[i]) {
            s[i] = fmax(tau0, tau0 - slack);
            t[i] = slack + s[i];
            lams[i] = mu0 / s[i];
            m_act++;
        } else {
            /* slack floor: tau0, but never more than a quarter of a two-sided hard constraint's width */
            double tau_c = tau0;
            const int other = (SGN(i) > 0) ? i + NC : i - NC;
            if (act[other] && !soft[other]) tau_c = fmin(tau0, 0.25 * (du[k * NC + c] - dl[k * NC + c]));
            t[i] = fmax(slack, tau_c);
        }
        lam[i] = mu0 / t[i];
        m_act++;
    }
    double res_g = 0, res_b = 0, res_d = 0, res_m = 0, mu = 0;
    for (it = 0;; it++) {
        /* ---- residuals ----
         * The stationarity and dynamics residuals are formed from the problem data at the first iterate; afterwards they FOLLOW THE STEP
         * (end of the loop): the Newton step solves the linearised rows exactly, so r_b <- (1 - alpha) r_b and
         * r_g <- (1 - alpha_d) r_g + (alpha - alpha_d) H dz -- no pass over A, B per iteration.  What the recursion cannot see is the
         * rounding error of the Riccati solve against the barrier-augmented Hessian (eps |H~| |dz|, which grows as the barrier weights
         * lam / t do): when the followed residuals pass the convergence test they are formed from the data again and tested again;
         * should that fail, the iteration goes on with residuals formed from the data in every iteration (exact_mode). */
        int exact = (it == 0) || exact_mode;
    recompute:
        res_g = res_b = res_d = res_m = 0; mu = 0;
        for (int k = 0; k < NS; k++) {
            for (int c = 0; c < NC; c++) {
                double rz = 0;
                for (int j = 0; j < NZ; j++) rz += R[(k * NC + c) * NZ + j] * z[k * NZ + j];
                Rz[k * NC + c] = rz;
            }
            if (!exact) {
                for (int j = 0; j < NZ; j++) res_g = fmax(res_g, fabs(rg[k * NZ + j]));
                if (k < N) for (int i = 0; i < NX; i++) res_b = fmax(res_b, fabs(rb[k * NX + i]));
                continue;
            }
            for (int j = 0; j < NZ; j++) {
                double acc = g[k * NZ + j];
                for (int l = 0; l < NZ; l++) acc += H[(k * NZ + j) * NZ + l] * z[k * NZ + l];
                if (k < N) {
                    for (int l = 0; l < NX; l++) {
                        double ab = (j < NX) ? A[(k * NX + l) * NX + j] : Bm[(k * NX + l) * NU + (j - NX)];
                        acc += ab * pi[(k + 1) * NX + l];
                    }
                }
                if (j < NX) acc -= pi[k * NX + j];
                for (int c = 0; c < NC; c++)
                    acc -= R[(k * NC + c) * NZ + j] * (lam[k * 2 * NC + c] - lam[k * 2 * NC + NC + c]);
                if (k == 0 && j < NX) acc = 0; /* x_0 is not a variable */
                if (k == N && j >= NX) acc = 0;
                rg[k * NZ + j] = acc;
                res_g = fmax(res_g, fabs(acc));
            }
            if (k < N)
                for (int i = 0; i < NX; i++) {
                    double acc = b[k * NX + i] - z[(k + 1) * NZ + i];
                    for (int l = 0; l < NX; l++) acc += A[(k * NX + i) * NX + l] * z[k * NZ + l];
                    for (int l = 0; l < NU; l++) acc += Bm[(k * NX + i) * NU + l] * z[k * NZ + NX + l];
                    rb[k * NX + i] = acc;
                    res_b = fmax(res_b, fabs(acc));
                }
        }
        for (int i = 0; i < NI; i++) {
            rd[i] = 0.0; rs[i] = 0.0;
            if (!act[i]) continue;
            rd[i] = SGN(i) * (Rz[ROW(i)] - bnd[i]) + s[i] - t[i];
            res_d = fmax(res_d, fabs(rd[i]));
            mu += lam[i] * t[i]; res_m = fmax(res_m, fabs(lam[i] * t[i]));
            if (soft[i]) {
                rs[i] = soft_Z[i] * s[i] + soft_z[i] - lam[i] - lams[i];
                res_g = fmax(res_g, fabs(rs[i]));
                mu += lams[i] * s[i]; res_m = fmax(res_m, fabs(lams[i] * s[i]));
            }
        }
        if (m_act > 0) mu /= m_act;
        if (orc_debug) fprintf(stderr, "ipm it %2d res_g %.3e res_b %.3e res_d %.3e res_m %.3e mu %.3e\n", it, res_g, res_b, res_d, res_m, mu);
        /* not-a-number in the QP's data (first pass over them): status 3 (reported as NaN, 1); a QP that diverges on the way: failed, status 4 */
        if (!(res_g == res_g) || !(res_b == res_b) || !(res_d == res_d) || !(res_m == res_m)) { status = (it == 0) ? 3 : 4; break; }
        {
            const int conv = res_g <= tol_g && res_b <= tol_b && res_d <= tol_d && res_m <= tol_m;
            if (conv && exact) { status = 0; break; }
            /* followed residuals that say converged are checked against the data -- and so are those the iteration limit stops at: the
             * loose acceptance below and the reported residuals are then values formed from A, B and R */
            if ((conv || it >= iter_max) && !exact) { exact = 1; exact_mode = 1; goto recompute; }
        }
        if (it >= iter_max) { status = 1; break; }

        /* ---- barrier-augmented Hessian ---- */
        for (int k = 0; k < NS; k++) memcpy(w.Ht[k], H + k * NZ * NZ, sizeof(double) * NZ * NZ);
        for (int i = 0; i < NI; i++) {
            if (!act[i]) continue;
            double gam = lam[i] / t[i];
            if (soft[i]) { const double gs = lams[i] / s[i]; gam = gam * (soft_Z[i] + gs) / (soft_Z[i] + gam + gs); }
            const int k = i / (2 * NC);
            const double *r = R + ROW(i) * NZ;
            for (int a2 = 0; a2 < NZ; a2++)
                for (int j = 0; j < NZ; j++) w.Ht[k][a2 * NZ + j] += gam * r[a2] * r[j];
        }
        /* ---- predictor (sigma = 0), then corrector ---- */
        /* separate step lengths for the primal (z, t, s) and the dual (pi, lam, lam_s) variables, as HPIPM's split_step */
        double alpha = 1.0, sigma = 0.0, alpha_d = 1.0;
        for (int pass = 0; pass < 2; pass++) {
            const double mu_target = fmax(sigma * mu, mu_floor); /* never aim below the tolerance */
            memcpy(gt, rg, sizeof(double) * NS * NZ);
            for (int i = 0; i < NI; i++) {
                if (!act[i]) continue;
                rm[i] = lam[i] * t[i]; rm2[i] = lams[i] * s[i];
                if (pass == 1) { rm[i] += dlam_a[i] * dt_a[i] - mu_target; if (soft[i]) rm2[i] += dlams_a[i] * ds_a[i] - mu_target; }
                double coef = (rm[i] + lam[i] * rd[i]) / t[i];
                if (soft[i]) {
                    const double gam = lam[i] / t[i], gs = lams[i] / s[i], D = soft_Z[i] + gam + gs;
                    coef -= gam * (rs[i] + coef + rm2[i] / s[i]) / D;
                }
                const int k = i / (2 * NC);
                const double *r = R + ROW(i) * NZ;
                for (int j = 0; j < NZ; j++) gt[k * NZ + j] += SGN(i) * coef * r[j];
            }
            riccati_backward(&w, pass == 0, gt, rb, p, kff);
            riccati_forward(&w, rb, p, kff, dz, dpi);
            /* slack / multiplier steps and the largest feasible step */
            double amax = 1.0, amax_d = 1.0;
            for (int k = 0; k < NS; k++)
                for (int c = 0; c < NC; c++) {
                    double drz = 0;
                    for (int j = 0; j < NZ; j++) drz += R[(k * NC + c) * NZ + j] * dz[k * NZ + j];
                    Rz[k * NC + c] = drz;   /* R dz */
                }
            for (int i = 0; i < NI; i++) {
                dt[i] = dlam[i] = ds[i] = dlams[i] = 0.0;
                if (!act[i]) continue;
                const double y = SGN(i) * Rz[ROW(i)];
                if (soft[i]) {
                    const double gam = lam[i] / t[i], gs = lams[i] / s[i], D = soft_Z[i] + gam + gs;
                    const double c1 = (rm[i] + lam[i] * rd[i]) / t[i], c2 = rm2[i] / s[i];
                    ds[i] = -(rs[i] + c1 + c2) / D - gam / D * y;
                    dlams[i] = -(rm2[i] + lams[i] * ds[i]) / s[i];
                    if (ds[i] < 0) amax = fmin(amax, -s[i] / ds[i]);
                    if (dlams[i] < 0) amax_d = fmin(amax_d, -lams[i] / dlams[i]);
                }
                dt[i] = y + ds[i] + rd[i];
                dlam[i] = -(rm[i] + lam[i] * dt[i]) / t[i];
                if (dt[i] < 0) amax = fmin(amax, -t[i] / dt[i]);
                if (dlam[i] < 0) amax_d = fmin(amax_d, -lam[i] / dlam[i]);
            }
            if (pass == 0) {
                double mu_aff = 0;
                for (int i = 0; i < NI; i++) {
                    if (!act[i]) continue;
                    mu_aff += (lam[i] + amax_d * dlam[i]) * (t[i] + amax * dt[i]);
                    if (soft[i]) mu_aff += (lams[i] + amax_d * dlams[i]) * (s[i] + amax * ds[i]);
                }
                if (m_act > 0) mu_aff /= m_act;
                double ratio = (mu > 0) ? mu_aff / mu : 0.0;
                sigma = ratio * ratio * ratio;
                memcpy(dlam_a, dlam, sizeof(double) * NI); memcpy(dt_a, dt, sizeof(double) * NI);
                memcpy(dlams_a, dlams, sizeof(double) * NI); memcpy(ds_a, ds, sizeof(double) * NI);
                if (m_act == 0) { alpha = alpha_d = 1.0; break; } /* no inequalities: Newton step is exact */
            } else {
                alpha = fmin(1.0, ORC_IPM_STEP_FRACTION * amax); alpha_d = fmin(1.0, ORC_IPM_STEP_FRACTION * amax_d);
            }
        }
        if (orc_debug) fprintf(stderr, "        sigma %.3e alpha %.6f alpha_d %.6f\n", sigma, alpha, alpha_d);
        if (fmin(alpha, alpha_d) < 1e-12) { status = 2; break; }
        for (int k = 0; k < NS; k++) {
            for (int j = 0; j < NZ; j++) z[k * NZ + j] += alpha * dz[k * NZ + j];
            for (int i = 0; i < NX; i++) pi[k * NX + i] += alpha_d * dpi[k * NX + i];
            /* residuals of the new iterate (see above); the rows of x_0 and of the padded terminal inputs stay zero */
            for (int j = 0; j < NZ; j++) {
                if ((k == 0 && j < NX) || (k == N && j >= NX)) continue;
                double hdz = 0;
                for (int l = 0; l < NZ; l++) hdz += H[(k * NZ + j) * NZ + l] * dz[k * NZ + l];
                rg[k * NZ + j] = (1.0 - alpha_d) * rg[k * NZ + j] + (alpha - alpha_d) * hdz;
            }
            if (k < N) for (int i = 0; i < NX; i++) rb[k * NX + i] *= (1.0 - alpha);
        }
        for (int i = 0; i < NI; i++)
            if (act[i]) {
                lam[i] += alpha_d * dlam[i]; t[i] += alpha * dt[i];
                if (soft[i]) { lams[i] += alpha_d * dlams[i]; s[i] += alpha * ds[i]; }
            }
    }
#undef SGN
#undef ROW
    /* max-iter exits that are converged to the loose tolerance (1e4 x tol) are reported as status 1
       ("acceptable": acados RTI tolerates ACADOS_MAXITER from the QP); otherwise status 4 = failed */
    if (status == 1 && !(res_g <= 1e4 * tol_g && res_b <= 1e4 * tol_b && res_d <= 1e4 * tol_d && res_m <= 1e4 * tol_m)) status = 4;
    if (sl) for (int i = 0; i < NI; i++) sl[i] = soft[i] ? s[i] : 0.0;
    if (stats) { stats[0] = res_g; stats[1] = res_b; stats[2] = res_d; stats[3] = res_m; stats[4] = mu; stats[5] = sg; stats[6] = sb; }
    if (iters) *iters = it;
    return status;
}

int orc_qp_solve(int N, const double *H, const double *g, const double *A, const double *Bm,
                 const double *b, const double *dx0, const double *R, const double *dl,
                 const double *du, int iter_max, double tol, double mu0, double tau0, double *z,
                 double *pi, double *lam, double *t, double *stats, int *iters)
{
    return orc_qp_solve_soft(N, H, g, A, Bm, b, dx0, R, dl, du, 0, 0, iter_max, tol, mu0, tau0, z, pi, lam, t, 0, stats, iters);
}
