// kernels_sqp.hip -- globalisation of the SQP mode: the backtracking line search on the l1 merit function that
// python/main.py:230-237 asks of acados (nlp_solver_type "SQP", nlp_solver_max_iter 2, globalization "MERIT_BACKTRACKING").
//
// One SQP iteration on the device is  [copy the iterate aside] -> k_linearize -> k_qp_wave (full step, QP multipliers)
// -> k_line_search, which per instance
//   * accepts the iterate it started from if its four KKT residuals meet the tolerances (status 0; later iterations leave it
//     alone), ends the solve on a QP failure (status 1 / 4), and reports 2 (ACADOS_MAXITER) when the iterations run out --
//     the reference accepts 0 and 2 (python/main.py:326);
//   * updates the merit weights from the QP multipliers: |mult| on the first iteration, then max(|mult|, (w + |mult|) / 2);
//   * evaluates  m(alpha) = cost + sum w_pi |x0 - x_0| + sum w_pi |Phi(x_k,u_k) - x_{k+1}| + sum w_lam max(0, violation)
//     along the QP step for alpha = 1, rho, rho^2, ... >= alpha_min and takes the first alpha with m(alpha) < m(0)
//     (or the Armijo condition m(alpha) - m(0) <= eps alpha D), alpha_min if none does;
//   * moves primal, slack and (unless full_step_dual) dual variables by alpha.
// One wavefront per instance, one lane per stage: a trial point costs one RK4 x M rollout of every interval WITHOUT
// sensitivities (about a fifth of a linearisation).  acados itself is absent and unpinned: this follows its documented
// behaviour, not its source.
#include "ihm2mpc_internal.h"
#include "model.hpp"
#include "sqp_body.hpp"

using namespace ihm2;

namespace {

template <int MODEL>
__global__ __launch_bounds__(64) void k_line_search(LsArgs a, int it, int last)
{
    if ((int)blockIdx.x >= a.B) return;
    line_search_body<MODEL>(a, blockIdx.x, it, last);
}

// the iterate the QP is built at, set aside for the line search: five arrays in one launch
__global__ __launch_bounds__(256) void k_copy_iterate(size_t n8, size_t n2, size_t nl, const double *__restrict__ x, const double *__restrict__ u,
                                                      const double *__restrict__ pi, const double *__restrict__ lam, const double *__restrict__ slk,
                                                      double *__restrict__ xp, double *__restrict__ up, double *__restrict__ pip, double *__restrict__ lamp,
                                                      double *__restrict__ slkp)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nl; e += stride) {
        lamp[e] = lam[e]; slkp[e] = slk[e];
        if (e < n8) { xp[e] = x[e]; pip[e] = pi[e]; }
        if (e < n2) up[e] = u[e];
    }
}

}  // namespace

void ihm2_launch_copy_iterate(ihm2mpc_handle *h)
{
    const size_t B = h->B, N = h->N, NS = h->NS;
    const size_t nl = B * NS * NLAM;          // the longest of the five
    const int blocks = (int)((nl + 255) / 256 < 4096 ? (nl + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_copy_iterate, dim3(blocks), dim3(256), 0, h->stream, B * NS * 8, B * N * 2, nl, h->x, h->u, h->pi, h->lam, h->slk, h->ls_x, h->ls_u, h->ls_pi,
                       h->ls_lam, h->ls_slk);
}

// it: SQP iteration index of this solve (0 resets the per-solve counters on the host side), last: it == max_iter - 1;
// phase / j_limit: LsArgs (the ladder in two launches for the collocation integrator)
void ihm2_launch_line_search(ihm2mpc_handle *h, int it, int last, int phase, int j_limit)
{
    LsArgs a = make_ls_args(h);
    a.phase = phase; a.j_limit = j_limit;
    if (h->cfg.model == IHM2MPC_MODEL_FDYN6U) hipLaunchKernelGGL(k_line_search<IHM2MPC_MODEL_FDYN6U>, dim3(h->B), dim3(64), 0, h->stream, a, it, last);
    else if (h->cfg.model == IHM2MPC_MODEL_FDYN6) hipLaunchKernelGGL(k_line_search<IHM2MPC_MODEL_FDYN6>, dim3(h->B), dim3(64), 0, h->stream, a, it, last);
    else hipLaunchKernelGGL(k_line_search<IHM2MPC_MODEL_FKIN6>, dim3(h->B), dim3(64), 0, h->stream, a, it, last);
}
