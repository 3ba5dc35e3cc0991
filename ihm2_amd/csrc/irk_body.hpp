// irk_body.hpp -- the collocation integrator's device code (see kernels_irk.hip for the mapping: four lanes per interval, Newton system by
// blocks), shared by the stand-alone IRK kernels and the persistent per-instance loop (kernels_qp.hip, k_steps).
#pragma once

#include "ihm2mpc_internal.h"
#include "device_steps.hpp"
#include "riccati_mfma.hpp"

namespace ihm2 {

struct IrkTab {
    double A[4][4], b[4];
    double invT[4][4], invD[4][4];      // (I + h/t_T A)^-1, (I + h/t_delta A)^-1
    double h;
};

// what a lane needs of the tableau: row st of A and of the two inverse actuator blocks, h and h b_st.  Extracted ONCE per kernel with
// constant indices: the kernel-argument struct must not have its address taken (the compiler would copy it to scratch)
struct IrkRows {
    double Arow[4], invT[4], invD[4], h, hb;
};
#define IRK_ROWS(R, tab, st)                                                                                                  \
    IrkRows R;                                                                                                                 \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++) {                                                                        \
        R.Arow[j_] = ((st) == 0) ? tab.A[0][j_] : ((st) == 1) ? tab.A[1][j_] : ((st) == 2) ? tab.A[2][j_] : tab.A[3][j_];           \
        R.invT[j_] = ((st) == 0) ? tab.invT[0][j_] : ((st) == 1) ? tab.invT[1][j_] : ((st) == 2) ? tab.invT[2][j_] : tab.invT[3][j_]; \
        R.invD[j_] = ((st) == 0) ? tab.invD[0][j_] : ((st) == 1) ? tab.invD[1][j_] : ((st) == 2) ? tab.invD[2][j_] : tab.invD[3][j_]; \
    }                                                                                                                          \
    R.h = tab.h;                                                                                                               \
    R.hb = tab.h * (((st) == 0) ? tab.b[0] : ((st) == 1) ? tab.b[1] : ((st) == 2) ? tab.b[2] : tab.b[3]);

// the same from a table in device memory (the persistent loop: a lane-dependent index is a plain load there)
__device__ __forceinline__ IrkRows irk_rows_from(const IrkTab *t, const int st)
{
    IrkRows r;
#pragma unroll
    for (int j = 0; j < 4; j++) { r.Arow[j] = t->A[st][j]; r.invT[j] = t->invT[st][j]; r.invD[j] = t->invD[st][j]; }
    r.h = t->h;
    r.hb = t->h * t->b[st];
    return r;
}

// row st of a 4 x 4 table that arrives as a kernel argument: selects with constant indices (a lane-dependent index would make the
// compiler copy the argument block to scratch)
__device__ __forceinline__ double tab_row(const double (&Tm)[4][4], int st, int j)
{
    return (st == 0) ? Tm[0][j] : (st == 1) ? Tm[1][j] : (st == 2) ? Tm[2][j] : Tm[3][j];
}

// value of lane j of this lane's quad
template <int J>
__device__ __forceinline__ double quad_bcast(double v)
{
    return dpp_mov<J | (J << 2) | (J << 4) | (J << 6)>(v);
}
__device__ __forceinline__ double quad_sum(double v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    return v;
}
__device__ __forceinline__ double quad_get(double v, int j)      // j is a compile-time constant after unrolling
{
    return (j == 0) ? quad_bcast<0>(v) : (j == 1) ? quad_bcast<1>(v) : (j == 2) ? quad_bcast<2>(v) : quad_bcast<3>(v);
}

// Gauss-Jordan on a 12 x 12 block distributed over the quad: lane i holds rows (i, 0..2) = Mr[3][12] and NR right-hand sides
// R[3][NR]; unknown (j, b) is column 3 j + b.  On return R holds the solution rows of this lane's stage.
template <int NR>
__device__ __forceinline__ void quad_solve12(const int st, double (&Mr)[3][12], double (&R)[3][NR])
{
    // (inner loops over all 12 columns with the condition c > p inside: trip counts that depend on p kept the compiler from
    // unrolling them, and the matrix went to scratch memory)
#pragma unroll
    for (int p = 0; p < 12; p++) {
        const int pj = p / 3, pb = p % 3;                 // the pivot row: local row pb of the lane that owns stage pj
        const bool owner = st == pj;
        const double ipiv = 1.0 / quad_get(Mr[pb][p], pj);
        double prow[12], prhs[NR];
#pragma unroll
        for (int c = 0; c < 12; c++) prow[c] = (c > p) ? quad_get(Mr[pb][c], pj) * ipiv : 0.0;
#pragma unroll
        for (int c = 0; c < NR; c++) prhs[c] = quad_get(R[pb][c], pj) * ipiv;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const bool is_piv = owner && r == pb;
            const double f = is_piv ? 0.0 : Mr[r][p];
#pragma unroll
            for (int c = 0; c < 12; c++)
                if (c > p) Mr[r][c] = is_piv ? prow[c] : fma(-f, prow[c], Mr[r][c]);
#pragma unroll
            for (int c = 0; c < NR; c++) R[r][c] = is_piv ? prhs[c] : fma(-f, prhs[c], R[r][c]);
        }
    }
}

// structural non-zeros of the model Jacobian (model.hpp: JX_MASK / JU_MASK): with constant indices the test folds at compile time, so
// the zero entries cost neither arithmetic nor registers (80 Jacobian entries per lane do not fit beside the 12 x 12 block)
template <int MODEL>
__device__ __forceinline__ constexpr bool jnz(int a, int c)
{
    return (c < 8) ? ((JX_MASK[MODEL ? 1 : 0][a] >> c) & 1u) != 0 : ((JU_MASK[MODEL ? 1 : 0][a] >> (c - 8)) & 1u) != 0;
}

template <int MODEL, bool ROW = false>
__device__ __forceinline__ void eval_model(const double (&X)[8], double u_T, double u_d, TrackSeg &trk, double (&f)[8], double (&J)[8][10])
{
    // (structural zeros of J are never written and never read: jnz)
    if (MODEL == IHM2MPC_MODEL_FKIN6) fkin6_eval<true>(X, u_T, u_d, trk, f, J);
    else fdyn6_eval<true, MODEL == IHM2MPC_MODEL_FDYN6U, ROW>(X, u_T, u_d, trk, f, J);
}

// One collocation step of size tab.h from x for the quad (st = this lane's stage).  K: this lane's stage value on return.
// With SENS: dK[a][c] = d K_st[a] / d (x, u)_c for the incoming sensitivity S (8 x 10, the same in the four lanes).
// state groups as compile-time index maps (tables in device memory would turn every J[a][G..] into a dynamic index and J into scratch)
#define G1(q) (6 + (q))
#define G2(q) (3 + (q))
#define G3(q) (q)

// Newton iterations of one collocation step from x; K: this lane's stage value, J: the model Jacobian at the final stage point
// (evaluated only WITH_J, for the sensitivities).
// ROW: the plants' variant (every quad of a 16-lane row on the same car): the dynamic model's wheels spread over the row (model.hpp)
template <int MODEL, bool WITH_J, bool ROW = false>
__device__ __forceinline__ void irk_step(const int st, const IrkRows &tab, const double (&x)[8], double u_T, double u_d, TrackSeg &trk,
                                         double (&K)[8], double (&J)[8][10])
{
    const double h = tab.h;
    const double (&Arow)[4] = tab.Arow, (&invT)[4] = tab.invT, (&invD)[4] = tab.invD;
#pragma unroll
    for (int a = 0; a < 8; a++) K[a] = 0.0;
    double f[8];
    for (int it = 0; it <= IHM2MPC_IRK_NEWTON_ITER; it++) {
        const bool last = it == IHM2MPC_IRK_NEWTON_ITER;      // the last pass only evaluates the Jacobians the sensitivities need
        if (last && !WITH_J) break;
        double X[8];
#pragma unroll
        for (int a = 0; a < 8; a++) {
            double acc = x[a];
#pragma unroll
            for (int j = 0; j < 4; j++) acc = fma(h * Arow[j], quad_get(K[a], j), acc);
            X[a] = acc;
        }
        eval_model<MODEL, ROW>(X, u_T, u_d, trk, f, J);
        if (last) break;
        // ---- Newton step: (I - h A (x) J) d = -(K - f), group by group ----
        double d[8];      // this lane's part of the step
        // actuators: constant 4 x 4 blocks
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int a = G1(q);
            const double r = -(K[a] - f[a]);
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) acc = fma((a == 6) ? invT[j] : invD[j], quad_get(r, j), acc);
            d[a] = acc;
        }
        // velocities, then pose: 12 x 12 blocks with the coupling to the groups already solved on the right-hand side
#pragma unroll
        for (int grp = 0; grp < 2; grp++) {
            double Mr[3][12], R[3][1];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int a = grp ? G3(r) : G2(r);
                double rhs = -(K[a] - f[a]);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    double cpl = 0.0;      // J[a][solved unknowns] . step of stage j
#pragma unroll
                    for (int q = 0; q < 2; q++) if (jnz<MODEL>(a, G1(q))) cpl = fma(J[a][G1(q)], quad_get(d[G1(q)], j), cpl);
                    if (grp)
#pragma unroll
                        for (int q = 0; q < 3; q++) if (jnz<MODEL>(a, G2(q))) cpl = fma(J[a][G2(q)], quad_get(d[G2(q)], j), cpl);
                    rhs = fma(h * Arow[j], cpl, rhs);
#pragma unroll
                    for (int bq = 0; bq < 3; bq++) {
                        const double unit = (j == st && bq == r) ? 1.0 : 0.0;
                        Mr[r][3 * j + bq] = jnz<MODEL>(a, grp ? G3(bq) : G2(bq)) ? unit - h * Arow[j] * J[a][grp ? G3(bq) : G2(bq)] : unit;
                    }
                }
                R[r][0] = rhs;
            }
            quad_solve12<1>(st, Mr, R);
#pragma unroll
            for (int r = 0; r < 3; r++) d[grp ? G3(r) : G2(r)] = R[r][0];
        }
#pragma unroll
        for (int a = 0; a < 8; a++) K[a] += d[a];
    }
}

// Sensitivity columns COL0 .. COL0 + NCOL - 1 of one collocation step that starts from S = [I 0] (the first and, for the reference's
// sim_method_num_steps = 1, only step): (I - h A (x) J) dK = [J_x | J_u], J at the final stage values.  dK[a][c]: this lane's stage.
template <int MODEL, int COL0, int NCOL>
__device__ __forceinline__ void irk_sens_cols(const int st, const IrkRows &tab, const double (&J)[8][10], double (&dK)[8][NCOL])
{
    const double h = tab.h;
    const double (&Arow)[4] = tab.Arow, (&invT)[4] = tab.invT, (&invD)[4] = tab.invD;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int a = G1(q);
#pragma unroll
        for (int c = 0; c < NCOL; c++) {
            double acc = 0.0;
            if (jnz<MODEL>(a, COL0 + c))
#pragma unroll
                for (int j = 0; j < 4; j++) acc = fma((a == 6) ? invT[j] : invD[j], quad_get(J[a][COL0 + c], j), acc);
            dK[a][c] = acc;
        }
    }
#pragma unroll
    for (int grp = 0; grp < 2; grp++) {
        double Mr[3][12], R[3][NCOL];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int a = grp ? G3(r) : G2(r);
#pragma unroll
            for (int c = 0; c < NCOL; c++) R[r][c] = jnz<MODEL>(a, COL0 + c) ? J[a][COL0 + c] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int c = 0; c < NCOL; c++) {
                    double cpl = 0.0;
#pragma unroll
                    for (int q = 0; q < 2; q++) if (jnz<MODEL>(a, G1(q))) cpl = fma(J[a][G1(q)], quad_get(dK[G1(q)][c], j), cpl);
                    if (grp)
#pragma unroll
                        for (int q = 0; q < 3; q++) if (jnz<MODEL>(a, G2(q))) cpl = fma(J[a][G2(q)], quad_get(dK[G2(q)][c], j), cpl);
                    R[r][c] = fma(h * Arow[j], cpl, R[r][c]);
                }
#pragma unroll
                for (int bq = 0; bq < 3; bq++) {
                    const double unit = (j == st && bq == r) ? 1.0 : 0.0;
                    Mr[r][3 * j + bq] = jnz<MODEL>(a, grp ? G3(bq) : G2(bq)) ? unit - h * Arow[j] * J[a][grp ? G3(bq) : G2(bq)] : unit;
                }
            }
        }
        quad_solve12<NCOL>(st, Mr, R);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < NCOL; c++) dK[grp ? G3(r) : G2(r)][c] = R[r][c];
    }
}


// one quad: interval k of instance b -> the record [A | B | b] as the RK4 kernels write it (live: the quad's results are stored; the
// quads past the end of a range repeat the last interval so that all lanes stay active for the DPP exchanges)
template <int MODEL>
__device__ __forceinline__ void irk_linearize_quad(const int st, const IrkRows &rows, const double *xk, const double *uk, const int tid, const int nknots,
                                                   const double *__restrict__ s_ref, const double *__restrict__ kappa_ref, double *rec, const bool live)
{
    double x[8];
#pragma unroll
    for (int a = 0; a < 8; a++) x[a] = xk[a];
    const double u_T = uk[0], u_d = uk[1];
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, x[0]);
    // one step (the reference's sim_method_num_steps = 1, python/main.py:236; ihm2mpc_create refuses IRK with M != 1): S starts
    // from [I 0], so the right-hand side of the sensitivity system is the Jacobian itself; the ten columns are solved five at a time
    // (register budget) and written out at once
    const double hb = rows.hb;
    double K[8], J[8][10];
    irk_step<MODEL, true>(st, rows, x, u_T, u_d, trk, K, J);
    {
        double dK[8][5];
        irk_sens_cols<MODEL, 0, 5>(st, rows, J, dK);
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const double v = ((a == c) ? 1.0 : 0.0) + quad_sum(hb * dK[a][c]);
                if (live && (a >> 1) == st) rec[a * 8 + c] = v;
            }
    }
    {
        double dK[8][5];
        irk_sens_cols<MODEL, 5, 5>(st, rows, J, dK);
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const int cc = 5 + c;
                const double v = ((a == cc) ? 1.0 : 0.0) + quad_sum(hb * dK[a][c]);
                if (live && (a >> 1) == st) { if (cc < 8) rec[a * 8 + cc] = v; else rec[64 + a * 2 + (cc - 8)] = v; }
            }
    }
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const double xn = x[a] + quad_sum(hb * K[a]);
        if (live && (a >> 1) == st) rec[80 + a] = xn - xk[8 + a];
    }
}

// one quad: Phi(xp_k + al (x_k - xp_k), up_k + al (u_k - up_k)) of one interval -> out (8), the line search's trial point
// Plant step of one instance by its quad (python/main.py:395-400,476-502): x <- IRK x M over M steps of the tableau's h; model -1 / -2: the
// kinematic / dynamic switch of python/main.py:482-489 (crossed / un-crossed slip angles), decided once at the start of the control period.
// Every lane of the quad ends with the new state.  Shared by k_sim_irk and the persistent loop (bit-identical plant steps).  The FOUR quads of a
// 16-lane row must be on the same car and active together: the dynamic model spreads its wheels over them (irk_step<.., ROW>).
__device__ __forceinline__ void irk_sim_quad(const int st, const IrkRows &rows, const int model, const int M, double (&x)[8], const double u_T, const double u_d,
                                             TrackSeg &trk)
{
    int mdl = model;
    if (model < 0) {
        const double beta = atan(k_rwd * tan(x[7]));
        const double v2 = x[3] * x[3] + x[4] * x[4];
        mdl = (v2 * sin(beta) / k_lR <= 3.0) ? IHM2MPC_MODEL_FKIN6 : (model == -2 ? IHM2MPC_MODEL_FDYN6U : IHM2MPC_MODEL_FDYN6);
    }
    const double hb = rows.hb;
    for (int m = 0; m < M; m++) {
        double K[8], J[8][10];
        // the model is the same for the four lanes of a quad; quads of a wave may differ (the branch re-converges per step)
        if (mdl == IHM2MPC_MODEL_FKIN6) irk_step<IHM2MPC_MODEL_FKIN6, false>(st, rows, x, u_T, u_d, trk, K, J);
        else if (mdl == IHM2MPC_MODEL_FDYN6) irk_step<IHM2MPC_MODEL_FDYN6, false, true>(st, rows, x, u_T, u_d, trk, K, J);
        else irk_step<IHM2MPC_MODEL_FDYN6U, false, true>(st, rows, x, u_T, u_d, trk, K, J);
#pragma unroll
        for (int a = 0; a < 8; a++) x[a] += quad_sum(hb * K[a]);
    }
}

template <int MODEL>
__device__ __forceinline__ void irk_rollout_quad(const int st, const IrkRows &rows, const int M, const double al, const double *x, const double *xp,
                                                 const double *u, const double *up, const int tid, const int nknots, const double *__restrict__ s_ref,
                                                 const double *__restrict__ kappa_ref, double *out, const bool live)
{
    double xs[8];
#pragma unroll
    for (int i = 0; i < 8; i++) xs[i] = xp[i] + al * (x[i] - xp[i]);
    const double u_T = up[0] + al * (u[0] - up[0]), u_d = up[1] + al * (u[1] - up[1]);
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, xs[0]);
    const double hb = rows.hb;
    for (int m = 0; m < M; m++) {
        double K[8], J[8][10];
        irk_step<MODEL, false>(st, rows, xs, u_T, u_d, trk, K, J);
#pragma unroll
        for (int i = 0; i < 8; i++) xs[i] += quad_sum(hb * K[i]);
    }
    if (live && st == 0)
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = xs[i];
}

}  // namespace ihm2
