// kernels_misc.hip -- layout conversion and the controller-side bookkeeping of one control step.
//
//  * AoS <-> SoA: the C-ABI speaks instance-major arrays (B, elems) like the reference's
//    solver.set/get (python/main.py:299-334); the device keeps [elem][Bp] (instance-minor).
//  * k_prepare: reference ramp + warm-start shift of IHM2Controller.compute_control
//    (python/main.py:303-322), fused, on device.
//  * k_init_guess: rollout of the model from x0 under Stanley-type feedback
//    (StanleyController.compute_control, python/main.py:139-163; torque: P-term only).
#include "ihm2mpc_internal.h"
#include "model.hpp"

using namespace ihm2;

namespace {

// 64x(elems) tile transpose through LDS: coalesced on both sides
__global__ __launch_bounds__(256) void k_aos_to_soa(int B, int Bp, int elems, const double *__restrict__ aos, double *__restrict__ soa)
{
    __shared__ double tile[64][65];
    const int b0 = blockIdx.x * 64, e0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    for (int r = ty; r < 64; r += 4) {                         // r: instance within tile, tx: element
        const int b = b0 + r, e = e0 + tx;
        tile[r][tx] = (b < B && e < elems) ? aos[(size_t)b * elems + e] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {                         // r: element within tile, tx: instance
        const int e = e0 + r, b = b0 + tx;
        if (e < elems && b < Bp) soa[(size_t)e * Bp + b] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void k_soa_to_aos(int B, int Bp, int elems, const double *__restrict__ soa, double *__restrict__ aos)
{
    __shared__ double tile[64][65];
    const int b0 = blockIdx.x * 64, e0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {                         // r: element, tx: instance
        const int e = e0 + r, b = b0 + tx;
        tile[r][tx] = (e < elems && b < Bp) ? soa[(size_t)e * Bp + b] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {                         // r: instance, tx: element
        const int b = b0 + r, e = e0 + tx;
        if (b < B && e < elems) aos[(size_t)b * elems + e] = tile[tx][r];
    }
}

__global__ void k_fill(size_t n, double *p, double v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// yref_j = [s0 + s_target j/N, 0 x 11], yref_e = [s0 + s_target, 0 x 7];
// x_j <- x_{j+1}, u_j <- u_{j+1} (j < N-1); x_{N-1} <- x_N; u_{N-1} <- 0   (python/main.py:303-322)
__global__ __launch_bounds__(64) void k_prepare(int B, int Bp, int N, double s_target, const double *__restrict__ x0,
                                                double *__restrict__ x, double *__restrict__ u,
                                                double *__restrict__ yref, double *__restrict__ yref_e)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const double s0 = x0[b];
    for (int j = 0; j < N; j++) {
        yref[(size_t)(j * NY) * Bp + b] = s0 + s_target * j / N;
#pragma unroll
        for (int i = 1; i < NY; i++) yref[(size_t)(j * NY + i) * Bp + b] = 0.0;
    }
    yref_e[b] = s0 + s_target;
#pragma unroll
    for (int i = 1; i < NX; i++) yref_e[(size_t)i * Bp + b] = 0.0;
    for (int j = 0; j < N - 1; j++) {
#pragma unroll
        for (int i = 0; i < NX; i++) x[(size_t)(j * NX + i) * Bp + b] = x[(size_t)((j + 1) * NX + i) * Bp + b];
#pragma unroll
        for (int i = 0; i < NU; i++) u[(size_t)(j * NU + i) * Bp + b] = u[(size_t)((j + 1) * NU + i) * Bp + b];
    }
#pragma unroll
    for (int i = 0; i < NX; i++) x[(size_t)((N - 1) * NX + i) * Bp + b] = x[(size_t)(N * NX + i) * Bp + b];
    u[(size_t)((N - 1) * NU + 0) * Bp + b] = 0.0;
    u[(size_t)((N - 1) * NU + 1) * Bp + b] = 0.0;
}

__global__ __launch_bounds__(64) void k_init_guess(int B, int Bp, int N, int M, double dt, double v_ref_scale, int nknots,
                                                   const double *__restrict__ s_ref, const double *__restrict__ kappa_ref,
                                                   const int32_t *__restrict__ track_id, const double *__restrict__ x0,
                                                   const double *__restrict__ lbu, const double *__restrict__ ubu,
                                                   const double *__restrict__ lg, const double *__restrict__ ug,
                                                   double *__restrict__ xs, double *__restrict__ us)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = x0[(size_t)i * Bp + b];
    const double v_ref = v_ref_scale * x[3];
    const int tid = track_id[b];
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, x[0]);
    const double h = dt / M;
    double J[8][10];
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) xs[(size_t)(k * 8 + i) * Bp + b] = x[i];
        // Stanley feedback (python/main.py:139-163), clipped to the input box and the rate rows
        double dk;
        const double kap = trk.kappa(x[0], dk);
        double u_T = 90.0 * (v_ref - x[3]);
        double arg = kap * k_lR;
        arg = fmin(fmax(arg, -0.9), 0.9);
        double u_d = atan(2.0 * tan(asin(arg))) - 1.8 * x[2] - atan(5.5 * x[1] / (2.0 + x[3]));
        u_T = fmin(fmax(u_T, fmax(lbu[k * 2 + 0], x[6] + lg[k * 2 + 0])), fmin(ubu[k * 2 + 0], x[6] + ug[k * 2 + 0]));
        u_d = fmin(fmax(u_d, fmax(lbu[k * 2 + 1], x[7] + lg[k * 2 + 1])), fmin(ubu[k * 2 + 1], x[7] + ug[k * 2 + 1]));
        us[(size_t)(k * 2 + 0) * Bp + b] = u_T;
        us[(size_t)(k * 2 + 1) * Bp + b] = u_d;
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
                fkin6_eval<false>(X, u_T, u_d, trk, K, J);
#pragma unroll
                for (int i = 0; i < 8; i++) xacc[i] = fma(wh, K[i], xacc[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = xacc[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) xs[(size_t)(N * 8 + i) * Bp + b] = x[i];
}

}  // namespace

void ihm2_launch_aos_to_soa(ihm2mpc_handle *h, const double *aos, double *soa, int elems)
{
    dim3 grid((h->Bp + 63) / 64, (elems + 63) / 64);
    hipLaunchKernelGGL(k_aos_to_soa, grid, dim3(256), 0, h->stream, h->B, h->Bp, elems, aos, soa);
}

void ihm2_launch_soa_to_aos(ihm2mpc_handle *h, const double *soa, double *aos, int elems)
{
    dim3 grid((h->Bp + 63) / 64, (elems + 63) / 64);
    hipLaunchKernelGGL(k_soa_to_aos, grid, dim3(256), 0, h->stream, h->B, h->Bp, elems, soa, aos);
}

void ihm2_launch_fill(ihm2mpc_handle *h, double *soa, int elems, double value)
{
    const size_t n = (size_t)elems * h->Bp;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, n, soa, value);
}

void ihm2_launch_prepare(ihm2mpc_handle *h, double s_target)
{
    hipLaunchKernelGGL(k_prepare, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, h->B, h->Bp, h->N, s_target, h->x0, h->x,
                       h->u, h->yref, h->yref_e);
}

void ihm2_launch_init_guess(ihm2mpc_handle *h, double v_ref_scale)
{
    hipLaunchKernelGGL(k_init_guess, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, h->B, h->Bp, h->N, h->cfg.M, h->cfg.dt,
                       v_ref_scale, h->cfg.nknots, h->s_ref, h->kappa_ref, h->track_id, h->x0, h->lbu, h->ubu, h->lg, h->ug,
                       h->x, h->u);
}
