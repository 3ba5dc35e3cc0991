// kernels_irk.hip -- acados' IRK integrator for this path: 4-stage collocation (Gauss-Legendre for the OCP, python/main.py:234-236;
// Radau IIA for the plants, python/main.py:395-400, python/sim.py:28-33), M steps per interval:
// IHM2MPC_IRK_NEWTON_ITER Newton iterations per step from K = 0 with a fresh Jacobian each (acados' defaults: sim_method_newton_iter
// 3, jac_reuse 0), forward sensitivities by the implicit-function theorem at the final stage values.
//
// Mapping: FOUR LANES PER (instance, interval) -- lane i of a quad owns collocation stage i: the four model + Jacobian evaluations
// of a Newton iteration run side by side, stage values travel by DPP quad broadcasts (VALU speed, no LDS).  The Newton matrix
// I - h (A (x) J) (32 x 32) is never formed: in the order (T, delta) -> (v_x, v_y, r) -> (s, n, psi) the model's Jacobian is block
// lower triangular, so the system falls into
//   * two scalar actuator blocks, 4 x 4 over the stages with CONSTANT coefficients (I + h/t A): inverted once on the host;
//   * two 12 x 12 blocks (3 states x 4 stages), each lane holding the three rows of its stage: Gauss-Jordan elimination with the
//     pivot row broadcast inside the quad -- 36 matrix entries + right-hand sides per lane, all in registers.
// No pivoting: the blocks are I - h a_ij J with |h a J| well below 1 for the non-stiff states (the stiff actuator lags are the
// constant blocks).  One shooting interval costs 3 x 4 + 4 model evaluations instead of RK4 x 25's 100.
#include "ihm2mpc_internal.h"
#include "device_steps.hpp"
#include "riccati_mfma.hpp"
#include "irk_tableaux.h"
#include "irk_body.hpp"

using namespace ihm2;

namespace {

// one quad: interval k of instance b.  Record [A | B | b] as the RK4 kernels write it.
template <int MODEL>
__global__ __launch_bounds__(64) void k_linearize_irk(int B, int N, int M, IrkTab tab, int nknots, const double *__restrict__ s_ref,
                                                       const double *__restrict__ kappa_ref, const int32_t *__restrict__ track_id, const double *xs,
                                                       const double *__restrict__ us, double *lin)
{
    const long total = (long)B * N;
    const long pr = min((long)blockIdx.x * 16 + (threadIdx.x >> 2), total - 1);      // the tail quads repeat the last pair (all lanes of a quad stay active for the DPP exchanges)
    const bool live = (long)blockIdx.x * 16 + (threadIdx.x >> 2) < total;
    const int st = threadIdx.x & 3;
    const int b = (int)(pr / N), k = (int)(pr % N);
    const double *xk = xs + ((size_t)b * (N + 1) + k) * 8, *uk = us + ((size_t)b * N + k) * 2;
    (void)M;
    IRK_ROWS(rows, tab, st)
    irk_linearize_quad<MODEL>(st, rows, xk, uk, track_id[b], nknots, s_ref, kappa_ref, lin + ((size_t)b * N + k) * LIN_REC, live);
}

// plant step: x_next = IRK x M over dt, SIXTEEN lanes per instance -- four quads (lane & 3 = collocation stage) that share the wheels of the dynamic
// model between them (irk_body.hpp: irk_sim_quad); model -1 / -2: the kin / dyn switch of python/main.py:482-489
__global__ __launch_bounds__(64) void k_sim_irk(int B, int model, int M, IrkTab tab, int nknots, const double *__restrict__ s_ref,
                                                const double *__restrict__ kappa_ref, const int32_t *__restrict__ track_id, const double *xs,
                                                const double *us, double *xn, const int32_t *__restrict__ active)
{
    const int bq = blockIdx.x * 4 + (threadIdx.x >> 4);
    const int b = min(bq, B - 1), st = threadIdx.x & 3;
    double x[8];
#pragma unroll
    for (int a = 0; a < 8; a++) x[a] = xs[(size_t)b * 8 + a];
    const double u_T = us[(size_t)b * 2], u_d = us[(size_t)b * 2 + 1];
    const int tid = track_id[b];
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, x[0]);
    const bool frozen = active && !active[b];
    IRK_ROWS(rows, tab, st)
    if (!frozen) irk_sim_quad(st, rows, model, M, x, u_T, u_d, trk);
    if (bq < B && (threadIdx.x & 15) == 0)
#pragma unroll
        for (int a = 0; a < 8; a++) xn[(size_t)b * 8 + a] = x[a];
}

// Trial points of the SQP mode's line search (sqp_body.hpp): Phi(xp_k + al (x_k - xp_k), up_k + al (u_k - up_k)) for EVERY step length of
// the backtracking ladder al_j = rho^j >= alpha_min and every interval, one quad each -> phi (n_alpha, B, N, 8).  The line search kernel
// has one lane per stage; the collocation step wants four, so its rollouts are done here, all trial points at once (an instance
// rarely needs more than the first: the rest is cheap insurance against a second launch per backtrack).
template <int MODEL>
__global__ __launch_bounds__(64) void k_rollout_irk(int B, int N, int M, int j_begin, int j_end, const int32_t *__restrict__ pending, double alpha_red, IrkTab tab, int nknots, const double *__restrict__ s_ref,
                                                    const double *__restrict__ kappa_ref, const int32_t *__restrict__ track_id, const double *__restrict__ x,
                                                    const double *__restrict__ u, const double *__restrict__ xp, const double *__restrict__ up, double *phi)
{
    // quads are numbered instance-major within a step length; pending: only the instances the first line-search launch left open
    const long total = (long)(j_end - j_begin) * B * N;
    const long q = (long)blockIdx.x * 16 + (threadIdx.x >> 2);
    const long pl = min(q, total - 1);
    const int st = threadIdx.x & 3;
    const int j = j_begin + (int)(pl / ((long)B * N));
    const long rem = pl - (long)(j - j_begin) * B * N;
    const int b = (int)(rem / N), k = (int)(rem % N);
    if (pending && !pending[b]) return;         // the whole quad leaves together
    const long pr = (long)j * B * N + rem;
    double al = 1.0;
    for (int jj = 0; jj < j; jj++) al *= alpha_red;          // the very products the line search forms
    const size_t ex = ((size_t)b * (N + 1) + k) * 8, eu = ((size_t)b * N + k) * 2;
    IRK_ROWS(rows, tab, st)
    irk_rollout_quad<MODEL>(st, rows, M, al, x + ex, xp + ex, u + eu, up + eu, track_id[b], nknots, s_ref, kappa_ref, phi + (size_t)pr * 8, q < total);
}

// host: tableau + the constant actuator blocks (I + h/t A)^-1 for the step size h
static void invert4(const double (&Min)[4][4], double (&inv)[4][4])
{
    double a[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { a[i][j] = Min[i][j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        for (int j = 0; j < 8; j++) { const double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        const double ip = 1.0 / a[c][c];
        for (int j = 0; j < 8; j++) a[c][j] *= ip;
        for (int r = 0; r < 4; r++) {
            if (r == c) continue;
            const double f = a[r][c];
            for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) inv[i][j] = a[i][4 + j];
}

static IrkTab make_tab(int type, double h)
{
    IrkTab t;
    const double (*A)[4] = (type == IHM2MPC_INTEG_IRK_RADAU4) ? IRK_RADAU4_A : IRK_GL4_A;
    const double *b = (type == IHM2MPC_INTEG_IRK_RADAU4) ? IRK_RADAU4_b : IRK_GL4_b;
    double MT[4][4], MD[4][4];
    for (int i = 0; i < 4; i++) {
        t.b[i] = b[i];
        for (int j = 0; j < 4; j++) {
            t.A[i][j] = A[i][j];
            MT[i][j] = ((i == j) ? 1.0 : 0.0) + h / k_tT * A[i][j];
            MD[i][j] = ((i == j) ? 1.0 : 0.0) + h / k_tdelta * A[i][j];
        }
    }
    invert4(MT, t.invT); invert4(MD, t.invD);
    t.h = h;
    return t;
}

}  // namespace

void ihm2_launch_linearize_irk(ihm2mpc_handle *h)
{
    const long total = (long)h->B * h->N;
    const int blocks = (int)((total + 15) / 16);
    const IrkTab tab = make_tab(h->cfg.integrator_type, h->cfg.dt / h->cfg.M);
#define LAUNCH_IRK(MD) hipLaunchKernelGGL(k_linearize_irk<MD>, dim3(blocks), dim3(64), 0, h->stream, h->B, h->N, h->cfg.M, tab, h->cfg.nknots, h->s_ref, \
                                          h->kappa_ref, h->track_id, h->x, h->u, h->lin)
    if (h->cfg.model == IHM2MPC_MODEL_FDYN6U) LAUNCH_IRK(IHM2MPC_MODEL_FDYN6U);
    else if (h->cfg.model == IHM2MPC_MODEL_FDYN6) LAUNCH_IRK(IHM2MPC_MODEL_FDYN6);
    else LAUNCH_IRK(IHM2MPC_MODEL_FKIN6);
#undef LAUNCH_IRK
}

void ihm2_launch_rollout_irk(ihm2mpc_handle *h, int j_begin, int j_end, double *phi, const int32_t *pending)
{
    const long total = (long)(j_end - j_begin) * h->B * h->N;
    const int blocks = (int)((total + 15) / 16);
    const IrkTab tab = make_tab(h->cfg.integrator_type, h->cfg.dt / h->cfg.M);
#define LAUNCH_RO(MD) hipLaunchKernelGGL(k_rollout_irk<MD>, dim3(blocks), dim3(64), 0, h->stream, h->B, h->N, h->cfg.M, j_begin, j_end, pending, h->sqp_alpha_red, tab, h->cfg.nknots, \
                                         h->s_ref, h->kappa_ref, h->track_id, h->x, h->u, h->ls_x, h->ls_u, phi)
    if (h->cfg.model == IHM2MPC_MODEL_FDYN6U) LAUNCH_RO(IHM2MPC_MODEL_FDYN6U);
    else if (h->cfg.model == IHM2MPC_MODEL_FDYN6) LAUNCH_RO(IHM2MPC_MODEL_FDYN6);
    else LAUNCH_RO(IHM2MPC_MODEL_FKIN6);
#undef LAUNCH_RO
}

int ihm2_upload_irk_tab(ihm2mpc_handle *h)
{
    if (h->cfg.integrator_type == IHM2MPC_INTEG_ERK) return 0;
    const IrkTab tab = make_tab(h->cfg.integrator_type, h->cfg.dt / h->cfg.M);
    if (!h->irk_tab && hipMalloc(&h->irk_tab, sizeof(IrkTab)) != hipSuccess) return 1;
    if (hipMemcpyAsync(h->irk_tab, &tab, sizeof(IrkTab), hipMemcpyHostToDevice, h->stream) != hipSuccess) return 1;
    return hipStreamSynchronize(h->stream) == hipSuccess ? 0 : 1;
}

// the plant's tableau for M_sim steps per control period, in device memory for the persistent loop (rebuilt when M_sim changes)
int ihm2_upload_sim_irk_tab(ihm2mpc_handle *h, int M_sim)
{
    if (h->cfg.sim_integrator_type == IHM2MPC_INTEG_ERK) return 1;
    if (h->sim_irk_tab && h->sim_irk_M == M_sim) return 0;
    const IrkTab tab = make_tab(h->cfg.sim_integrator_type, h->cfg.dt / M_sim);
    if (!h->sim_irk_tab && hipMalloc(&h->sim_irk_tab, sizeof(IrkTab)) != hipSuccess) return 1;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return 1;          // a launch in flight may still read the old tableau
    if (hipMemcpy(h->sim_irk_tab, &tab, sizeof(IrkTab), hipMemcpyHostToDevice) != hipSuccess) return 1;
    h->sim_irk_M = M_sim;
    return 0;
}

void ihm2_launch_sim_irk(ihm2mpc_handle *h, int model, int M_sim, const double *x, const double *u, double *xn, hipStream_t stream, const int32_t *active)
{
    const int blocks = (h->B + 3) / 4;
    const IrkTab tab = make_tab(h->cfg.sim_integrator_type, h->cfg.dt / M_sim);
    hipLaunchKernelGGL(k_sim_irk, dim3(blocks), dim3(64), 0, stream, h->B, model, M_sim, tab, h->cfg.nknots, h->s_ref, h->kappa_ref, h->track_id, x, u, xn, active);
}
