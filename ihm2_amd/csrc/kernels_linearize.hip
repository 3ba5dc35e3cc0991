// kernels_linearize.hip -- HOT LOOP 1 of the RTI step: for every (instance b, shooting interval k)
// integrate the model over dt with M classical RK4 sub-steps and propagate the forward sensitivities
// S = d x_m / d (x_k, u_k) through the same stages (exact derivative of the discrete map).
//
// Replaces what acados' ERK integrator does inside AcadosOcpSolver.solve() for the reference
// (python/main.py:325; options old/generate.py:23-25 with sim_method_num_steps = M).
// RK4 tableau as dpc/main.py:87-97.
//
// Mapping: one lane per (b, k) pair, k fastest (instance-major arrays): a wavefront reads 64
// consecutive 64-byte state rows (4 KB contiguous) and writes 64 consecutive 704-byte records.
// Sensitivities are held column-wise in registers; only the structurally non-zero entries of the
// 8x10 matrix are stored (52 for fkin6, 55 for fdyn6) and only the non-zeros of the model Jacobian (31 / 37) are multiplied
// (model.hpp: JX_MASK / S_COL_MASK).  Algorithmic traffic per pair: read 10 + 8 doubles, write 88.
#include <cstdlib>

#include "ihm2mpc_internal.h"
#include "model.hpp"
#include "device_steps.hpp"

using namespace ihm2;

namespace {

template <int MODEL>
__global__ __launch_bounds__(64) void k_linearize(
    int B, int N, int M, double dt, int nknots, const double *__restrict__ s_ref,
    const double *__restrict__ kappa_ref, const int32_t *__restrict__ track_id, const double *__restrict__ xs,
    const double *__restrict__ us, double *__restrict__ lin)
{
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    const int b = (int)(t / N);
    const int k = (int)(t % N);
    if (b >= B) return;
    extern __shared__ double s_lds[];
    dev_linearize<MODEL>(b, k, N, M, dt, nknots, s_ref, kappa_ref, track_id, xs, us, lin, s_lds + threadIdx.x);      // (unused by fkin6)
}

// ---- small batches (the single real-time controller, B = 1): one sensitivity COLUMN per lane ----
// With few instances the device is empty and the latency of a solve is what counts.  Here wave c of an interval block
// propagates only column c of S (10 waves per 64 intervals); every lane re-evaluates the model and its Jacobian (10x redundant,
// on otherwise idle SIMDs) but carries 8 instead of 52 sensitivity entries: 0.11 ms instead of 0.25 ms per linearisation.
// The formulas per column are those of dev_integrate_sens, but the compiler shares different subexpressions when only one
// column is needed, so the records agree with k_linearize's to rounding (1e-15 relative), not bit for bit; it is therefore used
// for up to 128 intervals only (B <= 3 at N = 40), where nothing is compared bit-wise with the batch path.
template <int COL>
__device__ __forceinline__ void dev_integrate_col_fkin6(const double *xk, const double *uk, int tid, int M, double dt, int nknots,
                                                       const double *__restrict__ s_ref, const double *__restrict__ kappa_ref, double *rec)
{
    constexpr int MODEL = IHM2MPC_MODEL_FKIN6;
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = xk[i];
    const double u_T = uk[0], u_d = uk[1];
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, x[0]);
    double S[8], Sacc[8], dK[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { S[i] = (COL == i) ? 1.0 : 0.0; dK[i] = 0.0; Sacc[i] = 0.0; }
    const double h = dt / M;
    for (int m = 0; m < M; m++) {
        double xacc[8], K[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { xacc[i] = x[i]; K[i] = 0.0; }
        sens_col_copy<MODEL, COL>(S, Sacc);
#pragma unroll 1
        for (int st = 0; st < 4; st++) {
            const double ah = (st == 0) ? 0.0 : ((st == 3) ? h : 0.5 * h);
            const double wh = (st == 0 || st == 3) ? h * (1.0 / 6.0) : h * (2.0 / 6.0);
            double X[8], J[8][10];
#pragma unroll
            for (int i = 0; i < 8; i++) X[i] = fma(ah, K[i], x[i]);
            fkin6_eval<true>(X, u_T, u_d, trk, K, J);
#pragma unroll
            for (int i = 0; i < 8; i++) xacc[i] = fma(wh, K[i], xacc[i]);
            sens_col_stage<MODEL, COL>(J, S, nullptr, Sacc, dK, ah, wh);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = xacc[i];
        sens_col_copy<MODEL, COL>(Sacc, S);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const double v = ((S_COL_MASK[0][COL] >> i) & 1u) ? S[i] : 0.0;
        if (COL < 8) rec[i * 8 + COL] = v; else rec[64 + i * 2 + (COL - 8)] = v;
        if (COL == 0) rec[80 + i] = x[i] - xk[8 + i];
    }
}

__global__ __launch_bounds__(64) void k_linearize_cols(int B, int N, int M, double dt, int nknots, const double *__restrict__ s_ref,
                                                       const double *__restrict__ kappa_ref, const int32_t *__restrict__ track_id,
                                                       const double *__restrict__ xs, const double *__restrict__ us, double *__restrict__ lin)
{
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    const int b = (int)(t / N), k = (int)(t % N);
    if (b >= B) return;
    const double *xk = xs + ((size_t)b * (N + 1) + k) * 8, *uk = us + ((size_t)b * N + k) * 2;
    double *rec = lin + ((size_t)b * N + k) * LIN_REC;
    const int tid = track_id[b];
    switch (blockIdx.y) {       // block-uniform: one column per wavefront
#define COL_CASE(c) case c: dev_integrate_col_fkin6<c>(xk, uk, tid, M, dt, nknots, s_ref, kappa_ref, rec); break;
    FOR_ALL_COLS(COL_CASE)
#undef COL_CASE
    }
}

// columns of the dynamic models' stage derivatives dK parked in LDS between the stages (k_linearize_dyn), and their positions there
#ifndef DK_PARK_FROM
#define DK_PARK_FROM 7
#endif
__host__ __device__ constexpr int dk_pos(int c, int i)
{
    int p = 0;
    for (int cc = DK_PARK_FROM; cc < 10; cc++)
        for (int ii = 0; ii < 8; ii++) {
            if (cc == c && ii == i) return p;
            if ((S_COL_MASK[1][cc] >> ii) & 1u) p++;
        }
    return p;
}
__host__ __device__ constexpr int dk_count() { return dk_pos(10, 0); }

// The dynamic models keep the integrator in the kernel itself (the code of dev_integrate_sens, written out): as a shared device
// function the compiler forwarded the LDS-parked base sensitivities through registers (+45 spill stores, +14 %); only the
// fkin6 integrator is shared with the persistent loop.
template <int MODEL>
__global__ __launch_bounds__(64) void k_linearize_dyn(
    int B, int N, int M, double dt, int nknots, const double *__restrict__ s_ref,
    const double *__restrict__ kappa_ref, const int32_t *__restrict__ track_id, const double *__restrict__ xs,
    const double *__restrict__ us, double *__restrict__ lin)
{
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    const int b = (int)(t / N);
    const int k = (int)(t % N);
    if (b >= B) return;

    const double *xk = xs + ((size_t)b * (N + 1) + k) * 8;
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = xk[i];
    const double u_T = us[((size_t)b * N + k) * 2 + 0];
    const double u_d = us[((size_t)b * N + k) * 2 + 1];
    const int tid = track_id[b];
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, x[0]);

    // S, Sacc, dK: [column][row]; only rows in S_COL_MASK[column] are ever touched
    constexpr bool S_IN_LDS = MODEL != IHM2MPC_MODEL_FKIN6;
    extern __shared__ double s_lds[];
    double *Sl = S_IN_LDS ? s_lds + threadIdx.x : nullptr;
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
    // The stage derivatives dK of the last columns (delta_0, u_T, u_delta: 21 entries) wait in LDS as well while the model is evaluated -- what the
    // 160 KB of a CU hold beside S at four waves: (55 + 21) x 512 B = 38 KB per wave.  They are fetched in front of the column's stage and put back
    // behind it: same arithmetic, fewer values alive across the forward-AD evaluation (scratch 400 -> see profiles/r4/kernel_resources.txt).
    double *Dl = S_IN_LDS ? Sl + s_count(1) * 64 : nullptr;
#pragma unroll
    for (int c = DK_PARK_FROM; c < 10; c++)
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (S_IN_LDS && ((S_COL_MASK[1][c] >> i) & 1u)) Dl[dk_pos(c, i) * 64] = 0.0;

    const double h = dt / M;
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
            else fdyn6_eval<true, MODEL == IHM2MPC_MODEL_FDYN6U, false, true>(X, u_T, u_d, trk, K, J);      // (with the scheduling fences between the wheels)
#pragma unroll
            for (int i = 0; i < 8; i++) xacc[i] = fma(wh, K[i], xacc[i]);
#define DYN_STAGE_COL(c)                                                                                                        \
            {                                                                                                                   \
                if (S_IN_LDS && c >= DK_PARK_FROM) {                                                                            \
                    _Pragma("unroll") for (int i = 0; i < 8; i++)                                                               \
                        if ((S_COL_MASK[1][c] >> i) & 1u) dK[c][i] = Dl[dk_pos(c, i) * 64];                                     \
                }                                                                                                               \
                sens_col_stage<MODEL, c>(J, S[c], Sl, Sacc[c], dK[c], ah, wh);                                                  \
                if (S_IN_LDS && c >= DK_PARK_FROM) {                                                                            \
                    _Pragma("unroll") for (int i = 0; i < 8; i++)                                                               \
                        if ((S_COL_MASK[1][c] >> i) & 1u) Dl[dk_pos(c, i) * 64] = dK[c][i];                                     \
                }                                                                                                               \
            }
            FOR_ALL_COLS(DYN_STAGE_COL)
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
    double *rec = lin + ((size_t)b * N + k) * LIN_REC;
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) rec[i * 8 + j] = ((S_COL_MASK[MODEL ? 1 : 0][j] >> i) & 1u) ? (S_IN_LDS ? Sacc[j][i] : S[j][i]) : 0.0;
#pragma unroll
        for (int j = 0; j < 2; j++) rec[64 + i * 2 + j] = ((S_COL_MASK[MODEL ? 1 : 0][8 + j] >> i) & 1u) ? (S_IN_LDS ? Sacc[8 + j][i] : S[8 + j][i]) : 0.0;
        rec[80 + i] = x[i] - xk[8 + i];
    }
}


// plant / rollout step: x_next = RK4 x M over dt, no sensitivities; model -1 (-2: with fdyn6u) = kin/dyn switch of
// python/main.py:482-489 (v^2 sin(beta) / l_R <= 3 -> kinematic, else dynamic)
__global__ __launch_bounds__(64) void k_sim_step(int B, int model, int M, double dt, int nknots,
                                                 const double *__restrict__ s_ref, const double *__restrict__ kappa_ref,
                                                 const int32_t *__restrict__ track_id, const double *xs,
                                                 const double *__restrict__ us, double *xn, const int32_t *__restrict__ active)
{
    // four lanes per instance (the wheels of the dynamic model, device_steps.hpp): a quad is either whole inside the batch or whole outside
    const int t = blockIdx.x * 64 + threadIdx.x, b = t >> 2;
    if (b >= B) return;
    dev_sim_step(b, t & 3, model, M, dt, nknots, s_ref, kappa_ref, track_id, xs, us, xn, active);
}

// the plain kinematic plant (model 0): the integrator of the shooting intervals on (x, u), see dev_sim_step_kin
__global__ __launch_bounds__(64) void k_sim_step_kin(int B, int M, double dt, int nknots, const double *__restrict__ s_ref,
                                                     const double *__restrict__ kappa_ref, const int32_t *__restrict__ track_id, const double *xs,
                                                     const double *__restrict__ us, double *xn, const int32_t *__restrict__ active, double *spare_rec)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    dev_sim_step_kin(b, M, dt, nknots, s_ref, kappa_ref, track_id, xs, us, xn, active, spare_rec);
}

}  // namespace

void ihm2_launch_linearize(ihm2mpc_handle *h)
{
    if (h->cfg.integrator_type != IHM2MPC_INTEG_ERK) { ihm2_launch_linearize_irk(h); return; }
    const long total = (long)h->B * h->N;
    const int blocks = (int)((total + 63) / 64);
    // diagnostic (tools/bench_linearize.py --cols): the column-parallel kernel -- one sensitivity column per wavefront, ten wavefronts
    // per block of 64 intervals -- at any batch size, to measure it against the lane-per-interval kernel
    static const bool force_cols = getenv("IHM2MPC_LINEARIZE_COLS") && getenv("IHM2MPC_LINEARIZE_COLS")[0] == '1';
    if (h->cfg.model == IHM2MPC_MODEL_FDYN6U)
        hipLaunchKernelGGL(k_linearize_dyn<IHM2MPC_MODEL_FDYN6U>, dim3(blocks), dim3(64), (s_count(1) + dk_count()) * 64 * sizeof(double), h->stream, h->B, h->N, h->cfg.M, h->cfg.dt,
                           h->cfg.nknots, h->s_ref, h->kappa_ref, h->track_id, h->x, h->u, h->lin);
    else if (h->cfg.model == IHM2MPC_MODEL_FDYN6)
        hipLaunchKernelGGL(k_linearize_dyn<IHM2MPC_MODEL_FDYN6>, dim3(blocks), dim3(64), (s_count(1) + dk_count()) * 64 * sizeof(double), h->stream, h->B, h->N, h->cfg.M, h->cfg.dt,
                           h->cfg.nknots, h->s_ref, h->kappa_ref, h->track_id, h->x, h->u, h->lin);
    else if (blocks <= 2 || force_cols)      // one or a few real-time controllers: the latency path (not bit-identical to the batch kernel, see above)
        hipLaunchKernelGGL(k_linearize_cols, dim3(blocks, 10), dim3(64), 0, h->stream, h->B, h->N, h->cfg.M, h->cfg.dt,
                           h->cfg.nknots, h->s_ref, h->kappa_ref, h->track_id, h->x, h->u, h->lin);
    else
        hipLaunchKernelGGL(k_linearize<IHM2MPC_MODEL_FKIN6>, dim3(blocks), dim3(64), 0, h->stream, h->B, h->N, h->cfg.M, h->cfg.dt,
                           h->cfg.nknots, h->s_ref, h->kappa_ref, h->track_id, h->x, h->u, h->lin);
}

void ihm2_launch_sim(ihm2mpc_handle *h, int model, int M_sim, const double *x, const double *u, double *xn, hipStream_t stream, const int32_t *active)
{
    if (h->cfg.sim_integrator_type != IHM2MPC_INTEG_ERK) { ihm2_launch_sim_irk(h, model, M_sim, x, u, xn, stream, active); return; }
    const int blocks = (h->B + 63) / 64;
    // the plain kinematic plant shares the integrator of the shooting intervals (bit-identical to lane N of the persistent loop); for
    // the few instances of a real-time controller (the batches that also take the column-parallel linearisation) its discarded
    // sensitivities would put 0.1 ms on the critical path of ihm2mpc_step: those take the state-only rollout, and so does a handle whose
    // shooting intervals use the collocation integrator (nothing to share: the loop runs the plant as a phase of its own on one lane)
    if (model == IHM2MPC_MODEL_FKIN6 && (long)h->B * h->N > 128 && h->cfg.integrator_type == IHM2MPC_INTEG_ERK)
        hipLaunchKernelGGL(k_sim_step_kin, dim3(blocks), dim3(64), 0, stream, h->B, M_sim, h->cfg.dt, h->cfg.nknots, h->s_ref, h->kappa_ref, h->track_id,
                           x, u, xn, active, h->lin + (size_t)h->B * h->N * LIN_REC);
    else        // four lanes per instance
        hipLaunchKernelGGL(k_sim_step, dim3((4 * h->B + 63) / 64), dim3(64), 0, stream, h->B, model, M_sim, h->cfg.dt,
                           h->cfg.nknots, h->s_ref, h->kappa_ref, h->track_id, x, u, xn, active);
}
