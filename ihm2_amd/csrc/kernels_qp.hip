// kernels_qp.hip -- HOT LOOPS 2 and 3 of the RTI step: Gauss-Newton QP assembly, the structured
// interior-point QP solve, the full step and the multiplier update -- what HPIPM does inside
// AcadosOcpSolver.solve() for the reference (python/main.py:228-233,325).
//
// QP per instance (k = 0..N, z_k = (dx_k, du_k)):
//   min  sum 1/2 z_k' H_k z_k + g_k' z_k     s.t.  dx_0 = x0 - x_0,  dx_{k+1} = A_k dx_k + B_k du_k + b_k,
//        lb - c(z) <= R_k z_k <= ub - c(z)   (8 state boxes, 2 input boxes, 2 general rows [C D])
// H_k = cost_scale * V'W_kV is constant and shared by the batch (python/mpc.py:49-64); g_k = H_k z_k - Gy_k yref_k.
// Solver: Mehrotra predictor-corrector primal-dual interior point; the Newton system is reduced to
// an equality-constrained LQ problem solved by a Riccati recursion (backward factor+vector sweep,
// forward sweep).  Tolerances are relative to sg = max(1,|g|_inf) and sb = max(1,|b|_inf,|dx0|_inf).
//
// v1 mapping: ONE LANE PER INSTANCE.  Every per-instance array is SoA [elem][Bp] in global memory so
// that a wavefront touches 64 consecutive doubles per access.  Per-stage 8x8/10x10 blocks live in
// registers with compile-time indices.  (DESIGN.md discusses the wave-per-instance successor.)
#include "ihm2mpc_internal.h"

namespace {

struct QpArgs {
    int B, Bp, N, iter_max;
    double tol, mu0, tau0;
    // shared
    const double *Hs, *Gy, *lbx, *ubx, *lbu, *ubu, *CD, *lg, *ug;
    // per instance
    double *x, *u;
    const double *x0, *yref, *yref_e;
    double *pi, *lam, *res, *u0;
    int32_t *status, *qp_iter;
    const double *A, *Bm, *bvec;
    double *g, *dl, *du, *z, *qpi, *qlam, *qt, *gt, *rb, *rd, *dz, *dpi, *dlam, *dt, *dlam_a, *dt_a, *P, *Gux, *Ginv, *p, *kff;
};

#define INF_BOUND 1e20
#define AT(arr, e) a.arr[(size_t)(e) * Bp + b]

__device__ __forceinline__ bool fin(double v) { return fabs(v) < INF_BOUND; }
__device__ __forceinline__ constexpr int sym8(int i, int j) { return (i <= j) ? (i * 8 - i * (i - 1) / 2 + (j - i)) : (j * 8 - j * (j - 1) / 2 + (i - j)); }
__device__ __forceinline__ constexpr int sym10(int i, int j) { return (i <= j) ? (i * 10 - i * (i - 1) / 2 + (j - i)) : (j * 10 - j * (j - 1) / 2 + (i - j)); }

// R_c . v for the 12 constraint rows of stage k: rows 0..9 are unit vectors, rows 10, 11 are [C D]_k
__device__ __forceinline__ void rows_times(const QpArgs &a, int k, int N, const double (&v)[10], double (&out)[12])
{
#pragma unroll
    for (int c = 0; c < 10; c++) out[c] = v[c];
    out[10] = 0.0; out[11] = 0.0;
    if (k < N) {
#pragma unroll
        for (int j = 0; j < 10; j++) {
            out[10] = fma(a.CD[(k * 2 + 0) * 10 + j], v[j], out[10]);
            out[11] = fma(a.CD[(k * 2 + 1) * 10 + j], v[j], out[11]);
        }
    }
}

__global__ __launch_bounds__(64) void k_qp_lane(QpArgs a)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    const int B = a.B, Bp = a.Bp, N = a.N, NS = N + 1;
    if (b >= B) return;

    // ------------------------------------------------------------------ QP data + NLP residuals
    double sg = 1.0, sb = 1.0;
    double r_stat = 0.0, r_eq = 0.0, r_ineq = 0.0, r_comp = 0.0;
    int m_act = 0;
    for (int k = 0; k <= N; k++) {
        double zk[10];
#pragma unroll
        for (int i = 0; i < 8; i++) zk[i] = AT(x, k * 8 + i);
        zk[8] = (k < N) ? AT(u, k * 2 + 0) : 0.0;
        zk[9] = (k < N) ? AT(u, k * 2 + 1) : 0.0;
        double yr[12];
#pragma unroll
        for (int i = 0; i < 12; i++) yr[i] = (k < N) ? AT(yref, k * 12 + i) : ((i < 8) ? AT(yref_e, i) : 0.0);
        double gk[10];
#pragma unroll
        for (int i = 0; i < 10; i++) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 10; j++) acc = fma(a.Hs[(k * 10 + i) * 10 + j], zk[j], acc);
#pragma unroll
            for (int j = 0; j < 12; j++) acc = fma(-a.Gy[(k * 10 + i) * 12 + j], yr[j], acc);
            gk[i] = acc;
            AT(g, k * 10 + i) = acc;
            if (i < 8 || k < N) sg = fmax(sg, fabs(acc));
        }
        // bounds relative to the current iterate
        double cz[12], dlk[12], duk[12];
        rows_times(a, k, N, zk, cz);
#pragma unroll
        for (int c = 0; c < 12; c++) {
            double lb = -INFINITY, ub = INFINITY;
            if (c < 8) { if (k >= 1) { lb = a.lbx[k * 8 + c]; ub = a.ubx[k * 8 + c]; } }
            else if (c < 10) { if (k < N) { lb = a.lbu[k * 2 + c - 8]; ub = a.ubu[k * 2 + c - 8]; } }
            else { if (k < N) { lb = a.lg[k * 2 + c - 10]; ub = a.ug[k * 2 + c - 10]; } }
            dlk[c] = fin(lb) ? lb - cz[c] : -INFINITY;
            duk[c] = fin(ub) ? ub - cz[c] : INFINITY;
            AT(dl, k * 12 + c) = dlk[c];
            AT(du, k * 12 + c) = duk[c];
            const double ll = AT(lam, k * 24 + c), lu = AT(lam, k * 24 + 12 + c);
            if (fin(dlk[c])) { m_act++; r_ineq = fmax(r_ineq, dlk[c]); r_comp = fmax(r_comp, fabs(ll * dlk[c])); }
            if (fin(duk[c])) { m_act++; r_ineq = fmax(r_ineq, -duk[c]); r_comp = fmax(r_comp, fabs(lu * duk[c])); }
        }
        // stationarity of the NLP Lagrangian with the incoming multipliers
        double st[10];
#pragma unroll
        for (int j = 0; j < 10; j++) st[j] = gk[j];
        if (k < N) {
#pragma unroll
            for (int l = 0; l < 8; l++) {
                const double pl = AT(pi, (k + 1) * 8 + l);
#pragma unroll
                for (int j = 0; j < 8; j++) st[j] = fma(AT(A, (k * 8 + l) * 8 + j), pl, st[j]);
                st[8] = fma(AT(Bm, (k * 8 + l) * 2 + 0), pl, st[8]);
                st[9] = fma(AT(Bm, (k * 8 + l) * 2 + 1), pl, st[9]);
                const double bl = AT(bvec, k * 8 + l);
                sb = fmax(sb, fabs(bl));
                r_eq = fmax(r_eq, fabs(bl));
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) st[j] -= AT(pi, k * 8 + j);
#pragma unroll
        for (int c = 0; c < 10; c++) st[c] -= AT(lam, k * 24 + c) - AT(lam, k * 24 + 12 + c);
        if (k < N) {
#pragma unroll
            for (int c = 10; c < 12; c++) {
                const double d = AT(lam, k * 24 + c) - AT(lam, k * 24 + 12 + c);
#pragma unroll
                for (int j = 0; j < 10; j++) st[j] = fma(-a.CD[(k * 2 + c - 10) * 10 + j], d, st[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 10; j++) {
            if (k == 0 && j < 8) continue;
            if (k == N && j >= 8) continue;
            r_stat = fmax(r_stat, fabs(st[j]));
        }
    }
    double dx0[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        dx0[i] = AT(x0, i) - AT(x, i);
        sb = fmax(sb, fabs(dx0[i]));
        r_eq = fmax(r_eq, fabs(dx0[i]));
    }
    AT(res, 0) = r_stat; AT(res, 1) = r_eq; AT(res, 2) = r_ineq; AT(res, 3) = r_comp;

    const double tol_g = a.tol * sg, tol_b = a.tol * sb, tol_d = a.tol * sb, tol_m = a.tol * sg;
    const double mu_floor = 0.1 * tol_m;
    const double mu0 = a.mu0 * sg;

    // ------------------------------------------------------------------ initial point
    for (int k = 0; k <= N; k++) {
        double zk[10];
#pragma unroll
        for (int i = 0; i < 10; i++) { zk[i] = (k == 0 && i < 8) ? dx0[i] : 0.0; AT(z, k * 10 + i) = zk[i]; }
#pragma unroll
        for (int i = 0; i < 8; i++) AT(qpi, k * 8 + i) = 0.0;
        double rz[12];
        rows_times(a, k, N, zk, rz);
#pragma unroll
        for (int c = 0; c < 12; c++) {
            const double l = AT(dl, k * 12 + c), uu = AT(du, k * 12 + c);
            const bool al = fin(l), au = fin(uu);
            double tau_c = a.tau0;
            if (al && au) tau_c = fmin(a.tau0, 0.25 * (uu - l));
            const double tl = al ? fmax(rz[c] - l, tau_c) : 1.0;
            const double tu = au ? fmax(uu - rz[c], tau_c) : 1.0;
            AT(qt, k * 24 + c) = tl; AT(qt, k * 24 + 12 + c) = tu;
            AT(qlam, k * 24 + c) = al ? mu0 / tl : 0.0;
            AT(qlam, k * 24 + 12 + c) = au ? mu0 / tu : 0.0;
        }
    }

    // ------------------------------------------------------------------ interior-point iterations
    int qstatus = 1, it = 0;
    double res_g = 0, res_b = 0, res_d = 0, res_m = 0, mu = 0;
    const double inv_m = (m_act > 0) ? 1.0 / m_act : 0.0;
    for (it = 0;; it++) {
        // ---- residuals: rg -> gt, rb, rd ----
        res_g = res_b = res_d = res_m = 0.0; mu = 0.0;
        for (int k = 0; k <= N; k++) {
            double zk[10], rz[12], rg[10];
#pragma unroll
            for (int i = 0; i < 10; i++) zk[i] = AT(z, k * 10 + i);
            rows_times(a, k, N, zk, rz);
#pragma unroll
            for (int j = 0; j < 10; j++) {
                double acc = AT(g, k * 10 + j);
#pragma unroll
                for (int l = 0; l < 10; l++) acc = fma(a.Hs[(k * 10 + j) * 10 + l], zk[l], acc);
                rg[j] = acc;
            }
            if (k < N) {
                double rbk[8];
#pragma unroll
                for (int i = 0; i < 8; i++) rbk[i] = AT(bvec, k * 8 + i) - AT(z, (k + 1) * 10 + i);
#pragma unroll
                for (int l = 0; l < 8; l++) {
                    const double pl = AT(qpi, (k + 1) * 8 + l);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const double alj = AT(A, (k * 8 + l) * 8 + j);
                        rg[j] = fma(alj, pl, rg[j]);
                        rbk[l] = fma(alj, zk[j], rbk[l]);
                    }
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const double blj = AT(Bm, (k * 8 + l) * 2 + j);
                        rg[8 + j] = fma(blj, pl, rg[8 + j]);
                        rbk[l] = fma(blj, zk[8 + j], rbk[l]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++) { AT(rb, k * 8 + i) = rbk[i]; res_b = fmax(res_b, fabs(rbk[i])); }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) rg[j] -= AT(qpi, k * 8 + j);
            double dlam_c[12];
#pragma unroll
            for (int c = 0; c < 12; c++) {
                const double l = AT(dl, k * 12 + c), uu = AT(du, k * 12 + c);
                const bool al = fin(l), au = fin(uu);
                const double ll = AT(qlam, k * 24 + c), lu = AT(qlam, k * 24 + 12 + c);
                const double tl = AT(qt, k * 24 + c), tu = AT(qt, k * 24 + 12 + c);
                dlam_c[c] = ll - lu;
                const double rdl = al ? (rz[c] - tl - l) : 0.0;
                const double rdu = au ? (uu - rz[c] - tu) : 0.0;
                AT(rd, k * 24 + c) = rdl; AT(rd, k * 24 + 12 + c) = rdu;
                res_d = fmax(res_d, fmax(fabs(rdl), fabs(rdu)));
                if (al) { mu += ll * tl; res_m = fmax(res_m, fabs(ll * tl)); }
                if (au) { mu += lu * tu; res_m = fmax(res_m, fabs(lu * tu)); }
            }
#pragma unroll
            for (int c = 0; c < 10; c++) rg[c] -= dlam_c[c];
            if (k < N) {
#pragma unroll
                for (int j = 0; j < 10; j++) {
                    rg[j] = fma(-a.CD[(k * 2 + 0) * 10 + j], dlam_c[10], rg[j]);
                    rg[j] = fma(-a.CD[(k * 2 + 1) * 10 + j], dlam_c[11], rg[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 10; j++) {
                if ((k == 0 && j < 8) || (k == N && j >= 8)) rg[j] = 0.0;
                AT(gt, k * 10 + j) = rg[j];
                res_g = fmax(res_g, fabs(rg[j]));
            }
        }
        mu *= inv_m;
        if (!(res_g == res_g) || !(res_b == res_b) || !(res_d == res_d) || !(res_m == res_m)) { qstatus = 3; break; }
        if (res_g <= tol_g && res_b <= tol_b && res_d <= tol_d && res_m <= tol_m) { qstatus = 0; break; }
        if (it >= a.iter_max) { qstatus = 1; break; }

        double alpha = 1.0, sigma = 0.0;
        for (int pass = 0; pass < 2; pass++) {
            // ---- backward Riccati sweep (pass 0: factor + vector; pass 1: vector only) ----
            double Pn[36], pn[8];
            for (int k = N; k >= 0; k--) {
                // modified gradient and barrier weights of this stage
                double gtk[10], gam[12];
#pragma unroll
                for (int j = 0; j < 10; j++) gtk[j] = AT(gt, k * 10 + j);
                double coef[12];
#pragma unroll
                for (int c = 0; c < 12; c++) {
                    const bool al = fin(AT(dl, k * 12 + c)), au = fin(AT(du, k * 12 + c));
                    const double ll = AT(qlam, k * 24 + c), lu = AT(qlam, k * 24 + 12 + c);
                    const double tl = AT(qt, k * 24 + c), tu = AT(qt, k * 24 + 12 + c);
                    gam[c] = (al ? ll / tl : 0.0) + (au ? lu / tu : 0.0);
                    double cf = 0.0;
                    if (pass == 0) {
                        if (al) cf += (ll * tl + ll * AT(rd, k * 24 + c)) / tl;
                        if (au) cf -= (lu * tu + lu * AT(rd, k * 24 + 12 + c)) / tu;
                    } else {
                        const double mu_t = fmax(sigma * mu, mu_floor);
                        if (al) cf += (AT(dlam_a, k * 24 + c) * AT(dt_a, k * 24 + c) - mu_t) / tl;
                        if (au) cf -= (AT(dlam_a, k * 24 + 12 + c) * AT(dt_a, k * 24 + 12 + c) - mu_t) / tu;
                    }
                    coef[c] = cf;
                }
#pragma unroll
                for (int c = 0; c < 10; c++) gtk[c] += coef[c];
                if (k < N) {
#pragma unroll
                    for (int j = 0; j < 10; j++) {
                        gtk[j] = fma(a.CD[(k * 2 + 0) * 10 + j], coef[10], gtk[j]);
                        gtk[j] = fma(a.CD[(k * 2 + 1) * 10 + j], coef[11], gtk[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 10; j++) AT(gt, k * 10 + j) = gtk[j];   // pass 1 adds its increment on top

                if (k == N) {
                    if (pass == 0) {
#pragma unroll
                        for (int i = 0; i < 8; i++)
#pragma unroll
                            for (int j = i; j < 8; j++) {
                                double v = a.Hs[(k * 10 + i) * 10 + j];
                                if (i == j) v += gam[i];
                                Pn[sym8(i, j)] = v;
                                AT(P, k * 36 + sym8(i, j)) = v;
                            }
                    } else {
#pragma unroll
                        for (int e = 0; e < 36; e++) Pn[e] = AT(P, k * 36 + e);
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) { pn[i] = gtk[i]; AT(p, k * 8 + i) = gtk[i]; }
                    continue;
                }
                // stage k < N
                double AB[8][10];
#pragma unroll
                for (int l = 0; l < 8; l++) {
#pragma unroll
                    for (int j = 0; j < 8; j++) AB[l][j] = AT(A, (k * 8 + l) * 8 + j);
                    AB[l][8] = AT(Bm, (k * 8 + l) * 2 + 0);
                    AB[l][9] = AT(Bm, (k * 8 + l) * 2 + 1);
                }
                // h = P_{k+1} rb_k + p_{k+1} ;  gz = gt_k + AB' h
                double hv[8], gz[10];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    double acc = pn[i];
#pragma unroll
                    for (int l = 0; l < 8; l++) acc = fma(Pn[sym8(i, l)], AT(rb, k * 8 + l), acc);
                    hv[i] = acc;
                }
#pragma unroll
                for (int j = 0; j < 10; j++) {
                    double acc = gtk[j];
#pragma unroll
                    for (int l = 0; l < 8; l++) acc = fma(AB[l][j], hv[l], acc);
                    gz[j] = acc;
                }
                double Gux[2][8], Gi[3];
                if (pass == 0) {
                    // G = Ht + AB' P AB, column by column (upper triangle)
                    double G[55];
#pragma unroll
                    for (int j = 0; j < 10; j++) {
                        double col[8];
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            double acc = 0.0;
#pragma unroll
                            for (int l = 0; l < 8; l++) acc = fma(Pn[sym8(i, l)], AB[l][j], acc);
                            col[i] = acc;
                        }
#pragma unroll
                        for (int i = 0; i <= j; i++) {
                            double acc = a.Hs[(k * 10 + i) * 10 + j];
                            if (i == j) acc += gam[j];
                            acc = fma(gam[10] * a.CD[(k * 2 + 0) * 10 + i], a.CD[(k * 2 + 0) * 10 + j], acc);
                            acc = fma(gam[11] * a.CD[(k * 2 + 1) * 10 + i], a.CD[(k * 2 + 1) * 10 + j], acc);
#pragma unroll
                            for (int l = 0; l < 8; l++) acc = fma(AB[l][i], col[l], acc);
                            G[sym10(i, j)] = acc;
                        }
                    }
                    const double g00 = G[sym10(8, 8)], g01 = G[sym10(8, 9)], g11 = G[sym10(9, 9)];
                    const double idet = 1.0 / (g00 * g11 - g01 * g01);
                    Gi[0] = g11 * idet; Gi[1] = -g01 * idet; Gi[2] = g00 * idet;
#pragma unroll
                    for (int j = 0; j < 8; j++) { Gux[0][j] = G[sym10(j, 8)]; Gux[1][j] = G[sym10(j, 9)]; }
#pragma unroll
                    for (int j = 0; j < 8; j++) { AT(Gux, k * 16 + j) = Gux[0][j]; AT(Gux, k * 16 + 8 + j) = Gux[1][j]; }
                    AT(Ginv, k * 3 + 0) = Gi[0]; AT(Ginv, k * 3 + 1) = Gi[1]; AT(Ginv, k * 3 + 2) = Gi[2];
                    // P_k = Gxx - Gux' Ginv Gux
                    double Kg[2][8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        Kg[0][j] = Gi[0] * Gux[0][j] + Gi[1] * Gux[1][j];
                        Kg[1][j] = Gi[1] * Gux[0][j] + Gi[2] * Gux[1][j];
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++)
#pragma unroll
                        for (int j = i; j < 8; j++) {
                            const double v = G[sym10(i, j)] - (Gux[0][i] * Kg[0][j] + Gux[1][i] * Kg[1][j]);
                            Pn[sym8(i, j)] = v;
                            AT(P, k * 36 + sym8(i, j)) = v;
                        }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++) { Gux[0][j] = AT(Gux, k * 16 + j); Gux[1][j] = AT(Gux, k * 16 + 8 + j); }
                    Gi[0] = AT(Ginv, k * 3 + 0); Gi[1] = AT(Ginv, k * 3 + 1); Gi[2] = AT(Ginv, k * 3 + 2);
#pragma unroll
                    for (int e = 0; e < 36; e++) Pn[e] = AT(P, k * 36 + e);
                }
                const double kf0 = Gi[0] * gz[8] + Gi[1] * gz[9], kf1 = Gi[1] * gz[8] + Gi[2] * gz[9];
                AT(kff, k * 2 + 0) = kf0; AT(kff, k * 2 + 1) = kf1;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    pn[i] = gz[i] - (Gux[0][i] * kf0 + Gux[1][i] * kf1);
                    AT(p, k * 8 + i) = pn[i];
                }
            }
            // ---- forward sweep + slack/multiplier steps + step length ----
            double dx[8];
#pragma unroll
            for (int i = 0; i < 8; i++) dx[i] = 0.0;
            double amax = 1.0;
            for (int k = 0; k <= N; k++) {
                double dzk[10];
#pragma unroll
                for (int i = 0; i < 8; i++) dzk[i] = dx[i];
                dzk[8] = 0.0; dzk[9] = 0.0;
                if (k < N) {
                    double t0 = 0.0, t1 = 0.0;
#pragma unroll
                    for (int j = 0; j < 8; j++) { t0 = fma(AT(Gux, k * 16 + j), dx[j], t0); t1 = fma(AT(Gux, k * 16 + 8 + j), dx[j], t1); }
                    const double g0 = AT(Ginv, k * 3 + 0), g1 = AT(Ginv, k * 3 + 1), g2 = AT(Ginv, k * 3 + 2);
                    dzk[8] = -(g0 * t0 + g1 * t1) - AT(kff, k * 2 + 0);
                    dzk[9] = -(g1 * t0 + g2 * t1) - AT(kff, k * 2 + 1);
                }
#pragma unroll
                for (int j = 0; j < 10; j++) AT(dz, k * 10 + j) = dzk[j];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    double acc = AT(p, k * 8 + i);
#pragma unroll
                    for (int l = 0; l < 8; l++) acc = fma(AT(P, k * 36 + sym8(i, l)), dx[l], acc);
                    AT(dpi, k * 8 + i) = acc;
                }
                // constraints of this stage
                double drz[12];
                rows_times(a, k, N, dzk, drz);
#pragma unroll
                for (int c = 0; c < 12; c++) {
                    const bool al = fin(AT(dl, k * 12 + c)), au = fin(AT(du, k * 12 + c));
                    const double mu_t = fmax(sigma * mu, mu_floor);
                    if (al) {
                        const double ll = AT(qlam, k * 24 + c), tl = AT(qt, k * 24 + c);
                        const double rm = (pass == 0) ? ll * tl : ll * tl + AT(dlam_a, k * 24 + c) * AT(dt_a, k * 24 + c) - mu_t;
                        const double dtl = drz[c] + AT(rd, k * 24 + c);
                        const double dll = -(rm + ll * dtl) / tl;
                        AT(dt, k * 24 + c) = dtl; AT(dlam, k * 24 + c) = dll;
                        if (dtl < 0.0) amax = fmin(amax, -tl / dtl);
                        if (dll < 0.0) amax = fmin(amax, -ll / dll);
                    } else { AT(dt, k * 24 + c) = 0.0; AT(dlam, k * 24 + c) = 0.0; }
                    if (au) {
                        const double lu = AT(qlam, k * 24 + 12 + c), tu = AT(qt, k * 24 + 12 + c);
                        const double rm = (pass == 0) ? lu * tu : lu * tu + AT(dlam_a, k * 24 + 12 + c) * AT(dt_a, k * 24 + 12 + c) - mu_t;
                        const double dtu = -drz[c] + AT(rd, k * 24 + 12 + c);
                        const double dlu = -(rm + lu * dtu) / tu;
                        AT(dt, k * 24 + 12 + c) = dtu; AT(dlam, k * 24 + 12 + c) = dlu;
                        if (dtu < 0.0) amax = fmin(amax, -tu / dtu);
                        if (dlu < 0.0) amax = fmin(amax, -lu / dlu);
                    } else { AT(dt, k * 24 + 12 + c) = 0.0; AT(dlam, k * 24 + 12 + c) = 0.0; }
                }
                if (k < N) {
                    double dxn[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        double acc = AT(rb, k * 8 + i);
#pragma unroll
                        for (int l = 0; l < 8; l++) acc = fma(AT(A, (k * 8 + i) * 8 + l), dx[l], acc);
                        acc = fma(AT(Bm, (k * 8 + i) * 2 + 0), dzk[8], acc);
                        acc = fma(AT(Bm, (k * 8 + i) * 2 + 1), dzk[9], acc);
                        dxn[i] = acc;
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) dx[i] = dxn[i];
                }
            }
            if (pass == 0) {
                if (m_act == 0) { alpha = 1.0; break; }
                double mu_aff = 0.0;
                for (int e = 0; e < NS * 24; e++) {
                    const int k = e / 24, c = e % 24;
                    const bool act = (c < 12) ? fin(AT(dl, k * 12 + c)) : fin(AT(du, k * 12 + c - 12));
                    const double dle = AT(dlam, e), dte = AT(dt, e);
                    AT(dlam_a, e) = dle; AT(dt_a, e) = dte;
                    if (act) mu_aff += (AT(qlam, e) + amax * dle) * (AT(qt, e) + amax * dte);
                }
                mu_aff *= inv_m;
                const double ratio = (mu > 0.0) ? mu_aff / mu : 0.0;
                sigma = ratio * ratio * ratio;
            } else {
                alpha = fmin(1.0, 0.995 * amax);
            }
        }
        if (alpha < 1e-12) { qstatus = 2; break; }
        for (int k = 0; k <= N; k++) {
#pragma unroll
            for (int j = 0; j < 10; j++) AT(z, k * 10 + j) += alpha * AT(dz, k * 10 + j);
#pragma unroll
            for (int i = 0; i < 8; i++) AT(qpi, k * 8 + i) += alpha * AT(dpi, k * 8 + i);
#pragma unroll
            for (int c = 0; c < 24; c++) {
                const bool act = (c < 12) ? fin(AT(dl, k * 12 + c)) : fin(AT(du, k * 12 + c - 12));
                if (act) { AT(qlam, k * 24 + c) += alpha * AT(dlam, k * 24 + c); AT(qt, k * 24 + c) += alpha * AT(dt, k * 24 + c); }
            }
        }
    }
    if (qstatus == 1 && !(res_g <= 1e4 * tol_g && res_b <= 1e4 * tol_b && res_d <= 1e4 * tol_d && res_m <= 1e4 * tol_m)) qstatus = 4;

    // ------------------------------------------------------------------ RTI update
    int st = 0;
    if (qstatus == 3) st = 1;
    else if (qstatus == 2 || qstatus == 4) st = 4;
    if (st == 0) {
        bool bad = false;
        for (int e = 0; e < NS * 10; e++) bad |= !isfinite(AT(z, e));
        if (bad) st = 1;
    }
    if (st == 0) {
        for (int k = 0; k <= N; k++) {
#pragma unroll
            for (int i = 0; i < 8; i++) AT(x, k * 8 + i) += AT(z, k * 10 + i);
            if (k < N) { AT(u, k * 2 + 0) += AT(z, k * 10 + 8); AT(u, k * 2 + 1) += AT(z, k * 10 + 9); }
#pragma unroll
            for (int i = 0; i < 8; i++) AT(pi, k * 8 + i) = (k == 0) ? 0.0 : AT(qpi, k * 8 + i);
#pragma unroll
            for (int c = 0; c < 24; c++) AT(lam, k * 24 + c) = AT(qlam, k * 24 + c);
        }
    }
    AT(u0, 0) = AT(u, 0);
    AT(u0, 1) = AT(u, 1);
    a.status[b] = st;
    a.qp_iter[b] = it;
}

}  // namespace

void ihm2_launch_qp(ihm2mpc_handle *h)
{
    QpArgs a;
    a.B = h->B; a.Bp = h->Bp; a.N = h->N; a.iter_max = h->cfg.ipm_iter_max;
    a.tol = h->cfg.ipm_tol; a.mu0 = h->cfg.ipm_mu0; a.tau0 = h->cfg.ipm_tau0;
    a.Hs = h->Hs; a.Gy = h->Gy; a.lbx = h->lbx; a.ubx = h->ubx; a.lbu = h->lbu; a.ubu = h->ubu; a.CD = h->CD; a.lg = h->lg; a.ug = h->ug;
    a.x = h->x; a.u = h->u; a.x0 = h->x0; a.yref = h->yref; a.yref_e = h->yref_e;
    a.pi = h->pi; a.lam = h->lam; a.res = h->res; a.u0 = h->u0; a.status = h->status; a.qp_iter = h->qp_iter;
    a.A = h->A; a.Bm = h->Bm; a.bvec = h->bvec;
    a.g = h->q_g; a.dl = h->q_dl; a.du = h->q_du; a.z = h->q_z; a.qpi = h->q_pi; a.qlam = h->q_lam; a.qt = h->q_t;
    a.gt = h->q_gt; a.rb = h->q_rb; a.rd = h->q_rd; a.dz = h->q_dz; a.dpi = h->q_dpi; a.dlam = h->q_dlam; a.dt = h->q_dt;
    a.dlam_a = h->q_dlam_a; a.dt_a = h->q_dt_a; a.P = h->q_P; a.Gux = h->q_Gux; a.Ginv = h->q_Ginv; a.p = h->q_p; a.kff = h->q_kff;
    hipLaunchKernelGGL(k_qp_lane, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, a);
}
