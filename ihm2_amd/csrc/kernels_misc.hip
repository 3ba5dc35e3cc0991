// kernels_misc.hip -- the controller-side bookkeeping of one control step, on device.
//
//  * k_prepare: reference ramp + warm-start shift of IHM2Controller.compute_control
//    (python/main.py:303-322), fused.
//  * k_init_guess: rollout of the model from x0 under Stanley-type feedback
//    (StanleyController.compute_control, python/main.py:139-163; torque: P-term only).
#include <algorithm>
#include <cmath>

#include "ihm2mpc_internal.h"
#include "model.hpp"
#include "device_steps.hpp"

using namespace ihm2;

namespace {

__global__ __launch_bounds__(64) void k_prepare(int B, int N, double s_target, int mode, const double *__restrict__ x0,
                                                double *__restrict__ x, double *__restrict__ u,
                                                double *__restrict__ yref, double *__restrict__ yref_e)
{
    if ((int)blockIdx.x >= B) return;
    dev_prepare(blockIdx.x, threadIdx.x, N, s_target, mode, x0, x, u, yref, yref_e);
}

__global__ __launch_bounds__(64) void k_wrap_lap(int B, int N, int nknots, const double *__restrict__ s_ref, const int32_t *__restrict__ track_id,
                                                 double *__restrict__ x0, double *__restrict__ x)
{
    if ((int)blockIdx.x >= B) return;
    dev_wrap_lap(blockIdx.x, threadIdx.x, N, nknots, s_ref, track_id, x0, x);
}

// MODEL: the model the rollout integrates -- the OCP's own model where that is usable as a simulator (fkin6, fdyn6u), the
// kinematic one for fdyn6 as written (open-loop unstable over the horizon).
// only_failed != nullptr: re-initialise only the instances whose last solve failed (status other than 0 and 2 = max-iter of
// the SQP mode, as python/main.py:326 accepts), and clear their multipliers.
template <int MODEL>
__global__ __launch_bounds__(64) void k_init_guess(int B, int N, int M, double dt, double v_ref_scale, int nknots,
                                                   const double *__restrict__ s_ref, const double *__restrict__ kappa_ref,
                                                   const int32_t *__restrict__ track_id, const double *__restrict__ x0,
                                                   const double *__restrict__ lbu, const double *__restrict__ ubu,
                                                   const double *__restrict__ lg, const double *__restrict__ ug,
                                                   double *__restrict__ xs, double *__restrict__ us,
                                                   const int32_t *__restrict__ only_failed, double *__restrict__ pi,
                                                   double *__restrict__ lam, int exact_lags)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    if (only_failed) {
        if (only_failed[b] == 0 || only_failed[b] == 2) return;
        for (int e = 0; e < (N + 1) * 8; e++) pi[(size_t)b * (N + 1) * 8 + e] = 0.0;
        for (int e = 0; e < (N + 1) * NLAM; e++) lam[(size_t)b * (N + 1) * NLAM + e] = 0.0;
    }
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = x0[(size_t)b * 8 + i];
    const double v_ref = v_ref_scale * x[3];
    const int tid = track_id[b];
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, x[0]);
    const double h = dt / M;
    // exact_lags (the recovery of failed instances, a guess the next RTI step corrects): the two actuator states are linear lags on a constant input
    // (python/models.py:251-252) -- taken in closed form at the stage times, so that the sub-step is chosen for the vehicle's dynamics (12.5 ms) and not
    // for the 1 ms torque lag that forces RK4 to 25 sub-steps per interval: a sixth of the model evaluations of a rollout the whole batch waits for
    const double eT2 = exact_lags ? exp(-0.5 * h / k_tT) : 1.0, eT = eT2 * eT2, eD2 = exact_lags ? exp(-0.5 * h / k_tdelta) : 1.0, eD = eD2 * eD2;
    double J[8][10];
    double *xb = xs + (size_t)b * (N + 1) * 8, *ub = us + (size_t)b * N * 2;
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) xb[k * 8 + i] = x[i];
        // Stanley feedback (python/main.py:139-163), clipped to the input box and the rate rows
        double dk;
        const double kap = trk.kappa(x[0], dk);
        double u_T = 90.0 * (v_ref - x[3]);
        double arg = kap * k_lR;
        arg = fmin(fmax(arg, -0.9), 0.9);
        double u_d = atan(2.0 * tan(asin(arg))) - 1.8 * x[2] - atan(5.5 * x[1] / (2.0 + x[3]));
        u_T = fmin(fmax(u_T, fmax(lbu[k * 2 + 0], x[6] + lg[k * 2 + 0])), fmin(ubu[k * 2 + 0], x[6] + ug[k * 2 + 0]));
        u_d = fmin(fmax(u_d, fmax(lbu[k * 2 + 1], x[7] + lg[k * 2 + 1])), fmin(ubu[k * 2 + 1], x[7] + ug[k * 2 + 1]));
        ub[k * 2 + 0] = u_T;
        ub[k * 2 + 1] = u_d;
        for (int m = 0; m < M; m++) {
            double xacc[8], K[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { xacc[i] = x[i]; K[i] = 0.0; }
            const double dT0 = x[6] - u_T, dD0 = x[7] - u_d;
#pragma unroll 1
            for (int st = 0; st < 4; st++) {
                const double ah = (st == 0) ? 0.0 : ((st == 3) ? h : 0.5 * h);
                const double wh = (st == 0 || st == 3) ? h * (1.0 / 6.0) : h * (2.0 / 6.0);
                double X[8];
#pragma unroll
                for (int i = 0; i < 8; i++) X[i] = fma(ah, K[i], x[i]);
                if (exact_lags) {
                    X[6] = fma(dT0, (st == 0) ? 1.0 : ((st == 3) ? eT : eT2), u_T);
                    X[7] = fma(dD0, (st == 0) ? 1.0 : ((st == 3) ? eD : eD2), u_d);
                }
                if (MODEL == IHM2MPC_MODEL_FDYN6U) fdyn6_eval<false, true>(X, u_T, u_d, trk, K, J);
                else fkin6_eval<false>(X, u_T, u_d, trk, K, J);
#pragma unroll
                for (int i = 0; i < 8; i++) xacc[i] = fma(wh, K[i], xacc[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = xacc[i];
            if (exact_lags) { x[6] = fma(dT0, eT, u_T); x[7] = fma(dD0, eD, u_d); }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) xb[N * 8 + i] = x[i];
}

}  // namespace

void ihm2_launch_wrap_lap(ihm2mpc_handle *h)
{
    hipLaunchKernelGGL(k_wrap_lap, dim3(h->B), dim3(64), 0, h->stream, h->B, h->N, h->cfg.nknots, h->s_ref, h->track_id, h->x0, h->x);
}

void ihm2_launch_prepare(ihm2mpc_handle *h, double s_target, int mode, hipStream_t stream)
{
    hipLaunchKernelGGL(k_prepare, dim3(h->B), dim3(64), 0, stream, h->B, h->N, s_target, mode, h->x0, h->x, h->u, h->yref,
                       h->yref_e);
}

void ihm2_launch_init_guess(ihm2mpc_handle *h, double v_ref_scale, int only_failed)
{
    const int32_t *mask = only_failed ? h->status : nullptr;
    // the rollout is an RK4 rollout whatever the OCP's integrator: with IRK (one step per interval) it takes the 25 sub-steps RK4 needs
    // on the actuator lags (a guess: the first linearisation sees its defects against the OCP's own discretisation)
    // (sub-steps of at most 2 ms: RK4 is stable on the 1 ms torque lag up to 2.78 ms -- a longer interval takes more of them)
    int M_roll = (h->cfg.integrator_type == IHM2MPC_INTEG_ERK) ? h->cfg.M : std::max(25, (int)std::ceil(h->cfg.dt / 2e-3));
    // recovery (ihm2mpc_reinit_failed; no counterpart in the reference, whose loop stops at the first bad status, python/main.py:326-328): the whole batch
    // waits for the rollouts of its few failed instances -- 3.3 ms of a 19-26 ms step of configs[2] with 25 sub-steps per interval -- so they take the
    // actuator lags in closed form and sub-steps of at most 12.5 ms (0.55 ms)
    const int exact_lags = only_failed ? 1 : 0;
    if (exact_lags) M_roll = std::max(4, (int)std::ceil(h->cfg.dt / 12.5e-3));
#define LAUNCH_IG(MD)                                                                                                          \
    hipLaunchKernelGGL(k_init_guess<MD>, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, h->B, h->N, M_roll, h->cfg.dt,      \
                       v_ref_scale, h->cfg.nknots, h->s_ref, h->kappa_ref, h->track_id, h->x0, h->lbu, h->ubu, h->lg, h->ug,    \
                       h->x, h->u, mask, h->pi, h->lam, exact_lags)
    // recovery of a few failed instances: the kinematic rollout is 5x cheaper and its defects are what one RTI step absorbs
    if (h->cfg.model == IHM2MPC_MODEL_FDYN6U && !only_failed) LAUNCH_IG(IHM2MPC_MODEL_FDYN6U);
    else LAUNCH_IG(IHM2MPC_MODEL_FKIN6);
#undef LAUNCH_IG
}
