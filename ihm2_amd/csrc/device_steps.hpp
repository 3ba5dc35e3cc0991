// device_steps.hpp -- the per-instance pieces of one control step as device functions, shared by the stand-alone kernels
// (kernels_misc.hip, kernels_linearize.hip) and the persistent per-instance loop (kernels_qp.hip, k_steps): the same source,
// so both paths produce the same bits.
//
//  * dev_prepare / dev_wrap_lap: reference ramp + warm-start shift of IHM2Controller.compute_control (python/main.py:303-322)
//    and the lap wrap; one wavefront per instance.
//  * dev_linearize: RK4 x M with forward sensitivities of one shooting interval (HOT LOOP 1); one lane per (instance, interval).
//  * dev_sim_step: the plant step (python/main.py:476-502); one lane per instance.
#pragma once

#include "ihm2mpc_internal.h"
#include "model.hpp"

namespace ihm2 {

// yref_j = [s0 + s_target j/N, 0 x 11], yref_e = [s0 + s_target, 0 x 7];
// x_j <- x_{j+1}, u_j <- u_{j+1} (j < N-1); x_{N-1} <- x_N; u_{N-1} <- 0   (python/main.py:303-322)
// One wavefront per instance; the old rows are read into registers before anything is overwritten.
__device__ __forceinline__ void dev_prepare(int b, int lane, int N, double s_target, int mode, const double *x0,
                                            double *x, double *u, double *yref, double *yref_e)
{
    double *xb = x + (size_t)b * (N + 1) * 8, *ub = u + (size_t)b * N * 2;
    if (mode & 1) {
        const double s0 = x0[(size_t)b * 8];
        double *yb = yref + (size_t)b * N * 12, *ye = yref_e + (size_t)b * 8;
        for (int e = lane; e < N * 12; e += 64) yb[e] = (e % 12 == 0) ? s0 + s_target * (e / 12) / N : 0.0;
        if (lane < 8) ye[lane] = (lane == 0) ? s0 + s_target : 0.0;
    }
    if (!(mode & 2)) return;
    // shift: element e of stage j takes the value of stage j+1 (j < N-1); stage N-1 takes stage N
    const int nx = N * 8;                    // rows 0..N-1 are rewritten, row N stays
    for (int base = 0; base < nx; base += 64) {
        const int e = base + lane;
        const double v = (e < nx) ? xb[e + 8] : 0.0;
        __syncthreads();                     // all reads of this chunk (incl. the overlap) before its writes
        if (e < nx) xb[e] = v;
        __syncthreads();
    }
    const int nu = N * 2;
    for (int base = 0; base < nu; base += 64) {
        const int e = base + lane;
        const double v = (e < nu - 2) ? ub[e + 2] : 0.0;      // u_{N-1} <- 0
        __syncthreads();
        if (e < nu) ub[e] = v;
        __syncthreads();
    }
}

// Lap wrap for closed loops that run longer than the track tables reach (three laps, s in [-L, 2L)): an instance whose car
// has passed s = L is moved back by one lap -- x0 and the s-component of its whole iterate -- which changes nothing physically
// (the tables are periodic) and keeps s inside the table for ever.  L = -s_ref[0] of the instance's track (Track::length).
__device__ __forceinline__ void dev_wrap_lap(int b, int lane, int N, int nknots, const double *__restrict__ s_ref, const int32_t *__restrict__ track_id,
                                             double *x0, double *x)
{
    const double L = -s_ref[(size_t)track_id[b] * nknots];
    if (!(x0[(size_t)b * 8] >= L)) return;            // wave-uniform
    for (int k = lane; k <= N; k += 64) x[((size_t)b * (N + 1) + k) * 8] -= L;
    if (lane == 0) x0[(size_t)b * 8] -= L;
}

// position of entry (column c, row i) among the structurally non-zero sensitivities (column-major, 52 / 55 entries)
__host__ __device__ constexpr int s_pos(int mdl, int c, int i)
{
    int p = 0;
    for (int cc = 0; cc < c; cc++)
        for (int b = 0; b < 8; b++) p += (S_COL_MASK[mdl][cc] >> b) & 1u;
    for (int b = 0; b < i; b++) p += (S_COL_MASK[mdl][c] >> b) & 1u;
    return p;
}
__host__ __device__ constexpr int s_count(int mdl) { return s_pos(mdl, 10, 0); }

// one RK4 stage of sensitivity column COL:  dX = S + ah*dK_prev ; dK = Jx dX + Ju[:,COL] ; Sacc += wh*dK
// SL != nullptr: the sub-step's base sensitivities S live in LDS (entry-major, one word per lane: Sl[pos * 64]) instead of
// registers -- the dynamic model's forward-AD evaluation needs the registers (it spilled 868 B per lane to scratch)
template <int MODEL, int COL>
__device__ __forceinline__ void sens_col_stage(const double (&J)[8][10], const double (&S)[8], const double *Sl, double (&Sacc)[8],
                                               double (&dK)[8], double ah, double wh)
{
    constexpr unsigned cm = S_COL_MASK[MODEL ? 1 : 0][COL];
    double dX[8];
#pragma unroll
    for (int l = 0; l < 8; l++)
        if ((cm >> l) & 1u) dX[l] = fma(ah, dK[l], Sl ? Sl[s_pos(MODEL ? 1 : 0, COL, l) * 64] : S[l]);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (!((cm >> i) & 1u)) continue;
        double acc = 0.0;
        if (COL >= 8 && ((JU_MASK[MODEL ? 1 : 0][i] >> (COL - 8)) & 1u)) acc = J[i][COL];
#pragma unroll
        for (int l = 0; l < 8; l++)
            if (((JX_MASK[MODEL ? 1 : 0][i] & cm) >> l) & 1u) acc = fma(J[i][l], dX[l], acc);
        dK[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
        if ((cm >> i) & 1u) Sacc[i] = fma(wh, dK[i], Sacc[i]);
}

template <int MODEL, int COL>
__device__ __forceinline__ void sens_col_copy(const double (&src)[8], double (&dst)[8])
{
    constexpr unsigned cm = S_COL_MASK[MODEL ? 1 : 0][COL];
#pragma unroll
    for (int i = 0; i < 8; i++)
        if ((cm >> i) & 1u) dst[i] = src[i];
}

#define FOR_ALL_COLS(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9)

// one lane: interval k of instance b.  Sl: this lane's column of the LDS copy of S (dynamic models), nullptr for fkin6
template <int MODEL>
// xk (8), uk (2): where to integrate from; x_next (8): the state the defect b is taken against; rec: the 88-double record
// [A | B | b] written at the end; xn_out (8) or nullptr: Phi(x_k, u_k) itself (the kinematic PLANT is this very function on
// (x0, u0): it then runs in lockstep with the interval lanes of the same wavefront, see k_steps)
__device__ __forceinline__ void dev_integrate_sens(
    const double *xk, const double *__restrict__ uk, const double *x_next, int tid, int M, double dt, int nknots, const double *__restrict__ s_ref,
    const double *__restrict__ kappa_ref, double *__restrict__ rec, double *xn_out, double *__restrict__ Sl)
{
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = xk[i];
    const double u_T = uk[0];
    const double u_d = uk[1];
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, x[0]);

    // S, Sacc, dK: [column][row]; only rows in S_COL_MASK[column] are ever touched
    constexpr bool S_IN_LDS = MODEL != IHM2MPC_MODEL_FKIN6;
    double S[10][8], Sacc[10][8], dK[10][8];
#pragma unroll
    for (int c = 0; c < 10; c++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            S[c][i] = (c == i) ? 1.0 : 0.0; dK[c][i] = 0.0;
            if (S_IN_LDS) {
                Sacc[c][i] = S[c][i];
                if ((S_COL_MASK[1][c] >> i) & 1u) Sl[s_pos(1, c, i) * 64] = S[c][i];
            }
        }

    const double h = dt / M;
    if (!S_IN_LDS) {
        // fkin6: per sub-step FIRST the four stage evaluations of the state (the stage points depend on the state only), their
        // Jacobians kept; THEN every sensitivity column through its four stages with the column's entries in registers throughout.
        // Stage-major order (all columns per stage) moved S, Sacc and dK -- 156 values that do not fit the 256 vector registers
        // next to the model evaluation -- between the accumulator file and the vector registers once per STAGE: 507 of the 1617
        // instructions of a stage were v_accvgpr moves; column-major order touches S once per SUB-STEP.  Same arithmetic per
        // column, bit-identical results.
        for (int m = 0; m < M; m++) {
            double xacc[8], K[8], J4[4][8][10];
#pragma unroll
            for (int i = 0; i < 8; i++) { xacc[i] = x[i]; K[i] = 0.0; }
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const double ah = (st == 0) ? 0.0 : ((st == 3) ? h : 0.5 * h);
                const double wh = (st == 0 || st == 3) ? h * (1.0 / 6.0) : h * (2.0 / 6.0);
                double X[8];
#pragma unroll
                for (int i = 0; i < 8; i++) X[i] = fma(ah, K[i], x[i]);
                fkin6_eval<true>(X, u_T, u_d, trk, K, J4[st]);
#pragma unroll
                for (int i = 0; i < 8; i++) xacc[i] = fma(wh, K[i], xacc[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = xacc[i];
#define SUBSTEP_COL(c)                                                                                                   \
            {                                                                                                            \
                double Sa[8], dKc[8];                                                                                    \
                _Pragma("unroll") for (int i = 0; i < 8; i++) { Sa[i] = S[c][i]; dKc[i] = 0.0; }                          \
                _Pragma("unroll") for (int st = 0; st < 4; st++) {                                                       \
                    const double ah = (st == 0) ? 0.0 : ((st == 3) ? h : 0.5 * h);                                       \
                    const double wh = (st == 0 || st == 3) ? h * (1.0 / 6.0) : h * (2.0 / 6.0);                          \
                    sens_col_stage<MODEL, c>(J4[st], S[c], nullptr, Sa, dKc, ah, wh);                                    \
                }                                                                                                        \
                sens_col_copy<MODEL, c>(Sa, S[c]);                                                                       \
            }
            FOR_ALL_COLS(SUBSTEP_COL)
#undef SUBSTEP_COL
        }
    } else
    for (int m = 0; m < M; m++) {
        double xacc[8], K[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { xacc[i] = x[i]; K[i] = 0.0; }
#define COPY_S_TO_ACC(c) sens_col_copy<MODEL, c>(S[c], Sacc[c]);
        if (!S_IN_LDS) { FOR_ALL_COLS(COPY_S_TO_ACC) }      // with S in LDS, Sacc already holds S from the previous sub-step
#pragma unroll 1
        for (int st = 0; st < 4; st++) {
            const double ah = (st == 0) ? 0.0 : ((st == 3) ? h : 0.5 * h);
            const double wh = (st == 0 || st == 3) ? h * (1.0 / 6.0) : h * (2.0 / 6.0);
            double X[8], J[8][10];
#pragma unroll
            for (int i = 0; i < 8; i++) X[i] = fma(ah, K[i], x[i]);
            if (MODEL == IHM2MPC_MODEL_FKIN6) fkin6_eval<true>(X, u_T, u_d, trk, K, J);
            else fdyn6_eval<true, MODEL == IHM2MPC_MODEL_FDYN6U>(X, u_T, u_d, trk, K, J);
#pragma unroll
            for (int i = 0; i < 8; i++) xacc[i] = fma(wh, K[i], xacc[i]);
#define STAGE_COL(c) sens_col_stage<MODEL, c>(J, S[c], S_IN_LDS ? Sl : nullptr, Sacc[c], dK[c], ah, wh);
            FOR_ALL_COLS(STAGE_COL)
        }
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = xacc[i];
#define COPY_ACC_TO_S(c) sens_col_copy<MODEL, c>(Sacc[c], S[c]);
        if (!S_IN_LDS) { FOR_ALL_COLS(COPY_ACC_TO_S) }
        else {
#pragma unroll
            for (int c = 0; c < 10; c++)
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if ((S_COL_MASK[1][c] >> i) & 1u) Sl[s_pos(1, c, i) * 64] = Sacc[c][i];
        }
    }

    // output record [A (8x8 row-major) | B (8x2) | b = Phi(x_k,u_k) - x_{k+1}]
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) rec[i * 8 + j] = ((S_COL_MASK[MODEL ? 1 : 0][j] >> i) & 1u) ? (S_IN_LDS ? Sacc[j][i] : S[j][i]) : 0.0;
#pragma unroll
        for (int j = 0; j < 2; j++) rec[64 + i * 2 + j] = ((S_COL_MASK[MODEL ? 1 : 0][8 + j] >> i) & 1u) ? (S_IN_LDS ? Sacc[8 + j][i] : S[8 + j][i]) : 0.0;
        rec[80 + i] = x[i] - x_next[i];
    }
    if (xn_out) {
#pragma unroll
        for (int i = 0; i < 8; i++) xn_out[i] = x[i];
    }
}

// one lane: interval k of instance b
template <int MODEL>
__device__ __forceinline__ void dev_linearize(
    int b, int k, int N, int M, double dt, int nknots, const double *__restrict__ s_ref,
    const double *__restrict__ kappa_ref, const int32_t *__restrict__ track_id, const double *xs,
    const double *us, double *lin, double *Sl)
{
    const double *xk = xs + ((size_t)b * (N + 1) + k) * 8;
    dev_integrate_sens<MODEL>(xk, us + ((size_t)b * N + k) * 2, xk + 8, track_id[b], M, dt, nknots, s_ref, kappa_ref,
                              lin + ((size_t)b * N + k) * LIN_REC, nullptr, Sl);
}

// plant / rollout step: x_next = RK4 x M over dt; model -1 (-2: with fdyn6u) = kin/dyn switch of
// python/main.py:482-489 (v^2 sin(beta) / l_R <= 3 -> kinematic, else dynamic).
// The plain kinematic plant (model 0, the OCP's own model) is NOT handled here but by dev_sim_step_kin below.
// FOUR lanes: instance b on the lanes q = 0..3 of a quad, all active -- every lane carries the state, the dynamic model's wheels are spread over
// the lanes (fdyn6_eval_quad: bit-identical to the one-lane evaluation), lane q = 0 stores the result
__device__ __forceinline__ void dev_sim_step(int b, int q, int model, int M, double dt, int nknots,
                                             const double *__restrict__ s_ref, const double *__restrict__ kappa_ref,
                                             const int32_t *__restrict__ track_id, const double *xs,
                                             const double *us, double *xn, const int32_t *active)
{
    if (active && !active[b]) {          // a frozen instance keeps its state (closed loops: failed or finished cars)
        if (xn != xs && q == 0) for (int i = 0; i < 8; i++) xn[(size_t)b * 8 + i] = xs[(size_t)b * 8 + i];
        return;
    }
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = xs[(size_t)b * 8 + i];
    const double u_T = us[(size_t)b * 2], u_d = us[(size_t)b * 2 + 1];
    const int tid = track_id[b];
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, x[0]);
    int mdl = model;
    if (model < 0) {
        const double beta = atan(k_rwd * tan(x[7]));
        const double v2 = x[3] * x[3] + x[4] * x[4];
        mdl = (v2 * sin(beta) / k_lR <= 3.0) ? IHM2MPC_MODEL_FKIN6 : (model == -2 ? IHM2MPC_MODEL_FDYN6U : IHM2MPC_MODEL_FDYN6);
    }
    const double h = dt / M;
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
            if (mdl == IHM2MPC_MODEL_FKIN6) fkin6_eval<false>(X, u_T, u_d, trk, K, J);
            else if (mdl == IHM2MPC_MODEL_FDYN6U) fdyn6_eval_quad<true>(q, X, u_T, u_d, trk, K);
            else fdyn6_eval_quad<false>(q, X, u_T, u_d, trk, K);
#pragma unroll
            for (int i = 0; i < 8; i++) xacc[i] = fma(wh, K[i], xacc[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = xacc[i];
    }
    if (q == 0)
#pragma unroll
        for (int i = 0; i < 8; i++) xn[(size_t)b * 8 + i] = x[i];
}


// The plain kinematic plant (model 0) is dev_integrate_sens on (x, u): the same arithmetic as a shooting interval, so that the
// persistent loop can run it on the spare lane of the linearisation for free (k_steps); its record goes to spare_rec
// (88 doubles per instance, never read).
__device__ __forceinline__ void dev_sim_step_kin(int b, int M, double dt, int nknots, const double *__restrict__ s_ref,
                                                 const double *__restrict__ kappa_ref, const int32_t *__restrict__ track_id,
                                                 const double *xs, const double *us, double *xn, const int32_t *active, double *spare_rec)
{
    if (active && !active[b]) {
        if (xn != xs) for (int i = 0; i < 8; i++) xn[(size_t)b * 8 + i] = xs[(size_t)b * 8 + i];
        return;
    }
    dev_integrate_sens<IHM2MPC_MODEL_FKIN6>(xs + (size_t)b * 8, us + (size_t)b * 2, xs + (size_t)b * 8, track_id[b], M, dt, nknots, s_ref, kappa_ref,
                                            spare_rec + (size_t)b * LIN_REC, xn + (size_t)b * 8, nullptr);
}

}  // namespace ihm2
