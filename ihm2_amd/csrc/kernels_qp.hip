// kernels_qp.hip -- HOT LOOPS 2 and 3 of the RTI step: Gauss-Newton QP assembly, the structured
// interior-point QP solve, the full step and the multiplier update -- what HPIPM does inside
// AcadosOcpSolver.solve() for the reference (python/main.py:228-233,325).
//
// QP per instance (k = 0..N, z_k = (dx_k, du_k)):
//   min  sum 1/2 z_k' H_k z_k + g_k' z_k     s.t.  dx_0 = x0 - x_0,  dx_{k+1} = A_k dx_k + B_k du_k + b_k,
//        lb - c(z) <= R_k z_k <= ub - c(z)   (8 state boxes, 2 input boxes, 2 general rows [C D], 2 track rows h;
//                                             any side may be soft: slack s >= 0 with cost z s + 1/2 Z s^2)
// H_k = cost_scale * V'W_kV is constant and shared by the batch (python/mpc.py:49-64); g_k = H_k z_k - Gy_k yref_k.
// Residuals: the stationarity and dynamics residuals are formed from the problem data (one pass over the records) at the first
// iterate; afterwards they FOLLOW THE STEP -- the Newton step solves the linearised rows exactly, so r_b <- (1 - alpha) r_b and
// r_g <- (1 - alpha_d) r_g + (alpha - alpha_d) H dz -- and are formed from the data once more when they pass the convergence test
// (the rounding error of the Riccati solve against the barrier-augmented Hessian is invisible to the recursion).
// Solver: Mehrotra predictor-corrector primal-dual interior point with separate primal / dual step lengths; the Newton
// system is reduced to an equality-constrained LQ problem solved by a Riccati recursion (soft sides: their slack block is
// eliminated per constraint first).  Tolerances are relative to sg = max(1,|g|_inf) and sb = max(1,|b|_inf,|dx0|_inf).
//
// Mapping: ONE WAVEFRONT PER INSTANCE (block = 64 lanes), 4 instances per CU (38.5 - 40.6 KB of LDS each).
//   * LDS: iterate z, pi, modified gradient, dynamics residual, Riccati vectors p_k, gains K_k, step dz; with UNI the
//     batch-shared stage Hessian and general rows.
//   * registers: multipliers / slacks / their steps -- each lane owns NSLOT constraint slots (320 two-sided slots at the
//     reference's dimensions = 5 per lane; soft sides are one-sided slots with 8 more registers each).
//   * HBM/L2: linearisation records [A|B|b] (704 B per stage) are STREAMED through LDS staging slots by a register ring
//     that runs four stages ahead of each sweep (one coalesced wave load per record); P_k and M_k = A - B K_k (512 B each)
//     are stored once per factorisation; M_k is streamed back by the vector recursion (transposed lane map) and the
//     forward recursion, P_k is read back in a fully parallel phase.
//   * the sequential recursions:
//       factor + predictor's vector recursion: riccati_mfma.hpp -- five v_mfma_f64_16x16x4_f64 per stage, operands and results
//                chained in registers (P_k, p_k, K_k, kff_k, M_k = A - B K_k and c_k = rb_k - B kff_k from the same products)
//       vector   p_k = gt_x - K'gt_u + M_k'(P_{k+1} rb_k + p_{k+1})   (corrector) } carried in REGISTERS: the 8-term contractions
//       forward  dx_{k+1} = c_k + M_k dx_k                                          } alternate between DPP (inside a group of 8
//     lanes) and DPP + v_permlane16/32_swap (across the groups), so a stage's result is laid out as the next one's operand;
//     du_k and dpi_k (and the corrector's kff_k) are recovered afterwards in parallel over all stages.
//   * 8x8 / 8x10 / 10x10 products: one output entry per lane, operands from LDS (row reads broadcast, column reads
//     consecutive: conflict-free); wave reductions by DPP / permlane butterflies (VALU speed, no LDS crossbar).
#include <cstring>

#include "ihm2mpc_internal.h"
#include "device_steps.hpp"
#include "sqp_body.hpp"
#include "irk_body.hpp"
#include "riccati_mfma.hpp"

using namespace ihm2;

namespace {

struct QpArgs {
    int B, N, iter_max, nslots, m_act;
    int nslots_can;     // entries of the 64-lane slot table: the order in which the slot sums (mu, mu_aff) are taken, whatever the number of waves per instance
    double tol, mu0, tau0;
    // shared
    const double *Hs, *Gy, *CD, *slot_lb, *slot_ub, *slot_zw, *slot_Zw;
    const int32_t *slot_kc;
    // per instance
    double *x, *u;
    const double *x0, *yref, *yref_e;
    double *pi, *lam, *res, *qp_res, *u0;
    int32_t *status, *qp_iter;
    const double *lin;
    double *g, *rg, *P, *M, *slk;
    // track rows
    const int32_t *track_id;
    const double *widths;
    double car_L, car_W;
    // lateral-acceleration row (PATH == 2): its multipliers and slack values, (B, N+1, 2) = lower, upper side -- beside the 28 columns of the other rows
    double *lam_a, *slk_a;
    int symmetrize;     // P_k := (P_k + P_k') / 2 in the factor sweep (riccati_mfma.hpp): needed by the open-loop unstable dynamic model as written (fdyn6)
};

#define INF_BOUND 1e20
// Depths of the prefetch rings: records of the factor sweep (stages ahead), rows of the vector / forward sweeps (a pass of those sweeps is
// 2 SWEEP_RING stages of straight-line code).  The kernel's loop is as large as the instruction cache two CUs share (k_qp_wave<5,0,0,1>:
// 77 KB of code, 64 KB of cache): unrolling is paid for in instruction fetches.
#ifndef RIC_RING
#define RIC_RING 4
#endif
#ifndef SWEEP_RING
#define SWEEP_RING 4
#endif
#define SWEEP_DL ((SWEEP_RING >= 8) ? 4 : 2)       // stages the LDS operands are fetched ahead

// The block is ONE wavefront: its lanes run in lockstep and the LDS serves one wave's instructions in
// order, so a hand-off through LDS needs no s_barrier -- only a compiler fence.  (A __syncthreads()
// would also drain vmcnt(0), i.e. stall every phase on the record prefetches and P_k stores in flight.)
#define WSYNC()                                                  \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

__device__ __forceinline__ bool fin(double v) { return fabs(v) < INF_BOUND; }

// Lane exchanges at VALU speed (no LDS crossbar): DPP moves inside a 16-lane row (dpp_mov, riccati_mfma.hpp), v_permlane16_swap /
// v_permlane32_swap (gfx950) across rows and halves.
// Butterfly over all 64 lanes: xor 1, xor 2 (quad_perm), mirror inside 8 (row_half_mirror), rotate by 8 inside 16 (row_ror:8),
// then the row and half swaps -- with both operands equal their two results are "mine" and "the partner's".
template <typename OP>
__device__ __forceinline__ double wave_reduce(double v, OP op)
{
    v = op(v, dpp_mov<0xB1>(v));
    v = op(v, dpp_mov<0x4E>(v));
    v = op(v, dpp_mov<0x141>(v));
    v = op(v, dpp_mov<0x128>(v));
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto a16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = op(__hiloint2double(b16[0], a16[0]), __hiloint2double(b16[1], a16[1]));
    lo = __double2loint(v); hi = __double2hiint(v);
    auto a32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return op(__hiloint2double(b32[0], a32[0]), __hiloint2double(b32[1], a32[1]));
}
// NaN-propagating max (fmax drops NaNs): used where a NaN must surface in the residual
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a || b != b) ? NAN : fmax(a, b); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce(v, [](double a, double b) { return fmax(a, b); }); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce(v, [](double a, double b) { return fmin(a, b); }); }
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce(v, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ double wave_nanmax(double v) { return wave_reduce(v, [](double a, double b) { return nanmax(a, b); }); }

// Sum over aligned groups of 8 consecutive lanes with DPP moves (VALU speed, no LDS crossbar):
// quad_perm(1,0,3,2), quad_perm(2,3,0,1), row_half_mirror.  Every lane of the group gets the total.
__device__ __forceinline__ double sum8(double v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    return v;
}

// Sum over the 8 lanes {w, w+8, ..., w+56} that share the position w within their group of 8: DPP row_ror:8 inside the 16-lane
// row, then the gfx950 row / half swaps (v_permlane16_swap, v_permlane32_swap: with both operands equal the two results are
// "mine" and "the partner's", so their sum is the butterfly step).  Every lane of the class gets the total; no LDS involved.
__device__ __forceinline__ double sum_stride8(double v)
{
    v += dpp_mov<0x128>(v);
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto a16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double(b16[0], a16[0]) + __hiloint2double(b16[1], a16[1]);
    lo = __double2loint(v); hi = __double2hiint(v);
    auto a32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b32[0], a32[0]) + __hiloint2double(b32[1], a32[1]);
}

// Ring of D coalesced 512-byte loads running D-1 stages ahead of a sequential sweep over a (N,64) array.
// DIR = -1: k = N-1 .. 0 ; DIR = +1: k = 0 .. N-1.  body(k, value of element `lane` of row k).
// The element a lane takes alternates with the step: e_even on steps 0, 2, ... and e_odd on steps 1, 3, ... (D is even).
// pre(k, odd, x0, x1) fetches the LDS operands of a step DL steps ahead of its use (the body's own LDS stores would otherwise
// pin every LDS load behind them -- the compiler cannot see that the addresses differ -- and put an LDS latency on the
// recursion chain);  body(k, value, odd, x0, x1).
template <int DIR, int D, int DL, typename P, typename F>
__device__ __forceinline__ void stream_rows(const double *rows, int N, int e_even, int e_odd, P &&pre, F &&body)
{
    static_assert(D % 2 == 0 && DL % 2 == 0 && D % DL == 0 && 3 * D <= QM_PAD, "the element index alternates with the step parity");
    // The prefetches run UNCLAMPED past the instance, by at most ceil(N / 2D) 2D + D - N < 3 D rows (D + DL stages of LDS
    // operands): the streamed array is padded by QM_PAD >= 3 D rows at both ends and the LDS operands sit inside the kernel's LDS carve-up with other arrays on both sides,
    // so every address is valid and the values fetched for stages outside [0, N) are never used.  (Clamping the indices
    // cost a third of the sweep's instructions in scalar min / shift / add chains.)
    // Two register sets used in turn (a stage of the first half of the loop body takes its row from set 0 and refills set 1, the second
    // half the other way round): every load writes a register whose last value is dead, so nothing has to be copied at the loop's back
    // edge.  (With ONE set the value in use and its refill were live together, the compiler rotated them with moves at the back edge, and
    // each move waited for its load: an s_waitcnt vmcnt(0) -- the whole ring drained -- every D stages.)
    double r[2][D], x0[DL], x1[DL];
    // Buffer loads: resource descriptor over the instance's rows and their padding (scalar), the row as scalar offset, the lane's element as
    // 32-bit vector offset -- no 64-bit address arithmetic per lane and load (three vector instructions per stage as global loads).
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(rows) - (size_t)QM_PAD * 64, 0, (N + 2 * QM_PAD) * 512, 0x00020000);
    const unsigned ve = (unsigned)e_even * 8u, vo = (unsigned)e_odd * 8u;
    typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
    auto load_row = [&](const int step, const unsigned voff) -> double {       // step: wave-uniform
        const int row = (DIR < 0) ? N - 1 - step : step;
        const u2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, (QM_PAD + row) * 512, 0);
        return __hiloint2double((int)v.y, (int)v.x);
    };
#pragma unroll
    for (int d = 0; d < D; d++) {
        r[0][d] = load_row(d, (d & 1) ? vo : ve);
        // the initial loads are issued in ring order (fence): the wait counts of the loop are the minimum over both ways into it, and a
        // reordered prologue (oldest slot loaded last) made the steady state wait for all but one load at the top of every pass
        __builtin_amdgcn_sched_barrier(0);
        if (d < DL) pre((DIR < 0) ? N - 1 - d : d, (d & 1) != 0, x0[d], x1[d]);
    }
    for (int s0 = 0; s0 < N; s0 += 2 * D) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int s = s0 + h * D + d;
                const int k = (DIR < 0) ? N - 1 - s : s;
                const double v = r[h][d], y0 = x0[d % DL], y1 = x1[d % DL];
                r[h ^ 1][d] = load_row(s + D, (d & 1) ? vo : ve);
                pre((DIR < 0) ? k - DL : k + DL, (d & 1) != 0, x0[d % DL], x1[d % DL]);
                if (s < N) body(k, v, (d & 1) != 0, y0, y1);
            }
        }
    }
}

// The sweeps' row streaming as it was before the diet below (global loads with per-lane 64-bit addresses): kept for the SQP instantiations of the
// persistent loop, whose register allocation the leaner form tips into scratch (live options: 449 k control steps/s with this form, 380 k with the other).
template <int DIR, int D, int DL, typename P, typename F>
__device__ __forceinline__ void stream_rows_v1(const double *rows, int N, int e_even, int e_odd, P &&pre, F &&body)
{
    static_assert(D % 2 == 0 && DL % 2 == 0 && D % DL == 0 && 3 * D <= QM_PAD, "the element index alternates with the step parity");
    // The prefetches run UNCLAMPED past the instance, by at most ceil(N / 2D) 2D + D - N < 3 D rows (D + DL stages of LDS
    // operands): the streamed array is padded by QM_PAD >= 3 D rows at both ends and the LDS operands sit inside the kernel's LDS carve-up with other arrays on both sides,
    // so every address is valid and the values fetched for stages outside [0, N) are never used.  (Clamping the indices
    // cost a third of the sweep's instructions in scalar min / shift / add chains.)
    // Two register sets used in turn (a stage of the first half of the loop body takes its row from set 0 and refills set 1, the second
    // half the other way round): every load writes a register whose last value is dead, so nothing has to be copied at the loop's back
    // edge.  (With ONE set the value in use and its refill were live together, the compiler rotated them with moves at the back edge, and
    // each move waited for its load: an s_waitcnt vmcnt(0) -- the whole ring drained -- every D stages.)
    double r[2][D], x0[DL], x1[DL];
    const double *p_even = rows + (size_t)((DIR < 0) ? N - 1 : 0) * 64 + e_even, *p_odd = rows + (size_t)((DIR < 0) ? N - 1 : 0) * 64 + e_odd;
#pragma unroll
    for (int d = 0; d < D; d++) {
        r[0][d] = ((d & 1) ? p_odd : p_even)[(ptrdiff_t)DIR * d * 64];
        // the initial loads are issued in ring order (fence): the wait counts of the loop are the minimum over both ways into it, and a
        // reordered prologue (oldest slot loaded last) made the steady state wait for all but one load at the top of every pass
        __builtin_amdgcn_sched_barrier(0);
        if (d < DL) pre((DIR < 0) ? N - 1 - d : d, (d & 1) != 0, x0[d], x1[d]);
    }
    for (int s0 = 0; s0 < N; s0 += 2 * D) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int s = s0 + h * D + d;
                const int k = (DIR < 0) ? N - 1 - s : s;
                const double v = r[h][d], y0 = x0[d % DL], y1 = x1[d % DL];
                r[h ^ 1][d] = ((d & 1) ? p_odd : p_even)[(ptrdiff_t)DIR * (s + D) * 64];
                pre((DIR < 0) ? k - DL : k + DL, (d & 1) != 0, x0[d % DL], x1[d % DL]);
                if (s < N) body(k, v, (d & 1) != 0, y0, y1);
            }
        }
    }
}

// The QP of instance b, solved by the calling wavefront (all 64 lanes, lane = threadIdx.x); sm: the block's dynamic LDS.
// Called by k_qp_wave (one launch per RTI iteration) and by the persistent per-instance loop k_steps.
// NW > 1 (k_qp_block, the latency kernel for batches smaller than the chip): NW wavefronts share the instance.  The phases that
// are parallel over stages / rows run over all NT = 64 NW threads, the constraint slots are spread over NT lanes (NSLOT is then the
// count per lane of THAT table), reductions go wave -> LDS -> block, hand-offs are s_barriers; the three sequential sweeps run on
// wave 0 while the others wait.  NW == 1 compiles to exactly the single-wave code (tid == lane, BSYNC == WSYNC).
template <int NSLOT, int NSOFT, int PATH, int UNI, int NW = 1, int LEAN = 1>
__device__ __forceinline__ void qp_wave_body(const QpArgs &a, const int b, double *sm, const bool want_res = true)
{
    // want_res = false (wave-uniform): the stationarity residual of the incoming iterate -- an output only (ihm2mpc_get_residuals), a third of
    // the QP set-up -- is skipped; the persistent RTI loop asks for it on its last step alone
    constexpr int NT = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#define BSYNC() do { if (NW == 1) WSYNC(); else __syncthreads(); } while (0)
    const int N = a.N, NS = N + 1;
    // constraint rows per stage held in LDS: 8 x boxes, 2 u boxes, 2 general rows (+ 2 track rows (+ the lateral-acceleration row, PATH == 2)); the
    // multiplier arrays in HBM always have the full NLAM = 28 columns (14 lower sides, then 14 upper sides); row 14 keeps its two in lam_a / slk_a
    constexpr int NCK = (PATH == 2) ? 15 : PATH ? 14 : 12;
    constexpr bool ALAT = PATH == 2;
    // UNI: the stage Hessians H_0..H_{N-1} and the general rows [C D]_k do not depend on k (the reference's OCP: one W, one C, D
    // for all stages, python/mpc.py:49-99).  They are then kept in LDS (H only where the budget of 40 KB per instance allows)
    // instead of being fetched through L2 with lane-dependent addresses in every phase.
    constexpr bool HL = UNI && !PATH, CL = UNI;
    // ---- LDS carve-up (doubles) ----
    double *z = sm;                  // NS*10  QP iterate
    double *gt = z + NS * 10;        // NS*10  stationarity residual / modified gradient
    double *pi = gt + NS * 10;       // NS*8   QP costates
    double *pv = pi + NS * 8;        // NS*8   Riccati vector p_k, then dpi_k
    double *rb = pv + NS * 8;        // N*8    dynamics residual
    double *gam = rb + N * 8;        // NS*NCK barrier weights per constraint slot
    double *cf = gam + NS * NCK;     // NS*NCK lam_l - lam_u, then gradient coefficients
    double *dz = cf + NS * NCK;      // NS*10  step
    double *kff = dz + NS * 10;      // N*4    feed-forward terms (2 used per stage)
    double *Kl = kff + N * 4;        // N*16   K_k = Guu^-1 Gux
    double *Ginv = Kl + N * 16;      // N*8    Guu^-1 as (Gi0, Gi1, Gi2, Gi1, 0, 0, 0, 0)
    double *Prb = Ginv + N * 8;      // N*8    P_{k+1} rb_k (same for predictor and corrector)
    double *tile = Prb + N * 8;      // 8*17   transpose tile of the factor sweep
    double *hc = tile + 136;         // NS*2   d h_R / d psi, d h_L / d psi of the track rows (PATH only)
    double *ha = hc + (PATH ? NS * 2 : 0);   // NS*4   d a_lat / d (v_x, v_y, T, delta) of the lateral-acceleration row (PATH == 2; zeros where the row is absent)
    double *Hl = ha + (ALAT ? NS * 4 : 0);   // 200  stage and terminal Hessian (HL only)
    double *CDl = Hl + (HL ? 200 : 0);       // 20   general rows (CL only)
    double *spv = CDl + (CL ? 20 : 0);       // 60   the (up to three) non-zeros of every row of the two Hessians (HL only) ...
    int *spc = reinterpret_cast<int *>(spv + 60);      // 60 ints: ... and their columns
#define HS(k, i, l) (HL ? Hl[(((k) == N) ? 100 : 0) + (i) * 10 + (l)] : a.Hs[((k) * 10 + (i)) * 10 + (l)])
#define CDV(k, r, j) (CL ? CDl[(r) * 10 + (j)] : a.CD[((k) * 2 + (r)) * 10 + (j)])

    // reductions over the instance's threads: wave butterfly, then (NW > 1) one LDS word per wave -- the transpose tile of the factor
    // sweep is free outside the sweep
    auto blk_reduce = [&](double v, auto op) -> double {
        v = wave_reduce(v, op);
        if (NW > 1) {
            __syncthreads();                    // earlier readers of the words are done
            if (lane == 0) tile[wv] = v;
            __syncthreads();
            v = tile[0];
#pragma unroll
            for (int w = 1; w < NW; w++) v = op(v, tile[w]);
        }
        return v;
    };
    auto blk_max = [&](double v) { return blk_reduce(v, [](double x, double y) { return fmax(x, y); }); };
    auto blk_min = [&](double v) { return blk_reduce(v, [](double x, double y) { return fmin(x, y); }); };
    auto blk_sum = [&](double v) { return blk_reduce(v, [](double x, double y) { return x + y; }); };
    auto blk_nanmax = [&](double v) { return blk_reduce(v, [](double x, double y) { return nanmax(x, y); }); };

    const double *xb = a.x + (size_t)b * NS * 8;
    const double *ub = a.u + (size_t)b * N * 2;
    const double *linb = a.lin + (size_t)b * N * LIN_REC;
    double *gb = a.g + (size_t)b * NS * 10;
    double *rgb = a.rg + (size_t)b * NS * 10;
    double *Pg = a.P + (size_t)b * NS * 64;
    double *Mg = a.M + (size_t)b * N * 64;
    double *pib = a.pi + (size_t)b * NS * 8;
    double *lamb = a.lam + (size_t)b * NS * 28;
    double *slkb = a.slk + (size_t)b * NS * 28;
    double *lamab = ALAT ? a.lam_a + (size_t)b * NS * 2 : nullptr;
    double *slkab = ALAT ? a.slk_a + (size_t)b * NS * 2 : nullptr;
    // column of the state a non-zero of the lateral-acceleration row sits in (v_x, v_y, T, delta), and back
    auto alat_slot = [](int j) -> int { return (j == 3) ? 0 : (j == 4) ? 1 : (j == 6) ? 2 : (j == 7) ? 3 : -1; };

    // ------------------------------------------------------------------ QP data + NLP residuals
    // gradient g_k = H_k z_k - Gy_k yref_k, and the stationarity of the NLP with the incoming multipliers
    if (HL) for (int e = tid; e < 200; e += NT) Hl[e] = a.Hs[(e < 100) ? e : N * 100 + (e - 100)];
    if (CL && tid < 20) CDl[tid] = a.CD[tid];
    if (UNI) BSYNC();
    // Rows of the batch-shared Hessians with at most three non-zeros each (the reference's cost y = [x; u; x_act - u], python/mpc.py:49-58,
    // couples an actuator state with its own input only): phase (i) of every iteration then multiplies the non-zeros alone, in column
    // order -- the same sums as the dense loop, whose other terms are exact zeros.
    bool h_sparse = false;
    if (HL) {
        int cnt = 0;
        if (tid < 20) {
#pragma unroll
            for (int l = 0; l < 10; l++) {
                const double v = Hl[tid * 10 + l];
                if (v != 0.0) { if (cnt < 3) { spv[tid * 3 + cnt] = v; spc[tid * 3 + cnt] = l; } cnt++; }
            }
            for (int q = cnt; q < 3; q++) { spv[tid * 3 + q] = 0.0; spc[tid * 3 + q] = 0; }
        }
        h_sparse = __ballot(cnt > 3) == 0ull;           // the 20 rows sit in wave 0
        if (NW > 1) h_sparse = blk_max((wv == 0 && !h_sparse) ? 1.0 : 0.0) == 0.0;
        BSYNC();
    }
    double sg = 1.0, sb = 1.0, r_stat = 0.0, r_eq = 0.0;
    double w_R = 0.0, w_L = 0.0;
    if (PATH) {
        // track rows (old/generate_acaods_interface.py:191-212) at the iterate, stages 1..N:
        //   h_R = n - L/2 sin|psi| + W/2 cos(psi) - w_R ,  h_L = -n + L/2 sin|psi| + W/2 cos(psi) - w_L
        // gradients (1, a_R) and (-1, a_L) in (n, psi); d|psi| = sign(psi), sign(0) = 0
        const int trk = a.track_id[b];
        w_R = a.widths[trk * 2 + 0]; w_L = a.widths[trk * 2 + 1];
        for (int k = tid; k < NS; k += NT) {
            const double psi = xb[k * 8 + 2], sgn = (psi > 0.0) - (psi < 0.0);
            const double dfoot = -0.5 * a.car_L * cos(fabs(psi)) * sgn, dlat = -0.5 * a.car_W * sin(psi);
            hc[k * 2 + 0] = (k >= 1) ? dfoot + dlat : 0.0;
            hc[k * 2 + 1] = (k >= 1) ? -dfoot + dlat : 0.0;
            if (ALAT) {
                // the lateral-acceleration row lives on the stages 1..N-1 (it is no terminal row: con_h_expr_e, old/generate_acaods_interface.py:209-212)
                double g4[4];
                alat_eval(xb[k * 8 + 3], xb[k * 8 + 4], xb[k * 8 + 6], xb[k * 8 + 7], g4);
#pragma unroll
                for (int q = 0; q < 4; q++) ha[k * 4 + q] = (k >= 1 && k < N) ? g4[q] : 0.0;
            }
        }
        BSYNC();
    }
    for (int e = tid; e < NS * 10; e += NT) {
        const int k = e / 10, j = e % 10;
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < 8; l++) acc = fma(HS(k, j, l), xb[k * 8 + l], acc);
        if (k < N) {
            acc = fma(HS(k, j, 8), ub[k * 2 + 0], acc);
            acc = fma(HS(k, j, 9), ub[k * 2 + 1], acc);
            const double *yr = a.yref + ((size_t)b * N + k) * 12;
#pragma unroll
            for (int l = 0; l < 12; l++) acc = fma(-a.Gy[(k * 10 + j) * 12 + l], yr[l], acc);
        } else {
            const double *yr = a.yref_e + (size_t)b * 8;
#pragma unroll
            for (int l = 0; l < 8; l++) acc = fma(-a.Gy[(k * 10 + j) * 12 + l], yr[l], acc);
        }
        gb[e] = acc;
        const bool counted = !((k == 0 && j < 8) || (k == N && j >= 8));
        if (j < 8 || k < N) sg = fmax(sg, fabs(acc));
        // stationarity: g + AB' pi_{k+1} - [pi_k;0] - R'(lam_l - lam_u)
        if (!want_res) continue;
        double st = acc;
        if (k < N) {
            const double *rec = linb + (size_t)k * LIN_REC;
#pragma unroll
            for (int l = 0; l < 8; l++) st = fma((j < 8) ? rec[l * 8 + j] : rec[64 + l * 2 + (j - 8)], pib[(k + 1) * 8 + l], st);
        }
        if (j < 8) st -= pib[k * 8 + j];
        st -= lamb[k * 28 + j] - lamb[k * 28 + 14 + j];
        if (k < N) {
            st = fma(-CDV(k, 0, j), lamb[k * 28 + 10] - lamb[k * 28 + 24], st);
            st = fma(-CDV(k, 1, j), lamb[k * 28 + 11] - lamb[k * 28 + 25], st);
        }
        if (PATH && (j == 1 || j == 2)) {
            const double l12 = lamb[k * 28 + 12] - lamb[k * 28 + 26], l13 = lamb[k * 28 + 13] - lamb[k * 28 + 27];
            st -= (j == 1) ? l12 - l13 : hc[k * 2] * l12 + hc[k * 2 + 1] * l13;
        }
        if (ALAT && alat_slot(j) >= 0) st -= ha[k * 4 + alat_slot(j)] * (lamab[k * 2] - lamab[k * 2 + 1]);
        if (counted) r_stat = fmax(r_stat, fabs(st));
    }
    for (int e = tid; e < N * 8; e += NT) {
        const double bl = linb[(size_t)(e / 8) * LIN_REC + 80 + (e % 8)];
        sb = fmax(sb, fabs(bl));
        r_eq = fmax(r_eq, fabs(bl));
    }
    if (tid < 8) {
        const double d = a.x0[(size_t)b * 8 + tid] - xb[tid];
        sb = fmax(sb, fabs(d));
        r_eq = fmax(r_eq, fabs(d));
    }
    sg = blk_max(sg); sb = blk_max(sb);

    // constraint slots owned by this lane
    int s_kc[NSLOT];
    double s_dl[NSLOT], s_du[NSLOT];      // bounds relative to the iterate (+-inf = absent)
    double lam_l[NSLOT], lam_u[NSLOT], t_l[NSLOT], t_u[NSLOT];
    // soft slots (one-sided by construction, always among the first NSOFT of a lane): slack variable s >= 0 with cost
    // zw s + 1/2 Zw s^2 and multiplier lam_s; Zw < 0 marks a hard slot
    constexpr int NSO = (NSOFT > 0) ? NSOFT : 1;
    double so_zw[NSO], so_Zw[NSO], so_s[NSO], so_ls[NSO], so_rs[NSO], so_ds[NSO], so_dls[NSO], so_pa[NSO];
#pragma unroll
    for (int r = 0; r < NSO; r++) { so_zw[r] = 0.0; so_Zw[r] = -1.0; so_s[r] = 1.0; so_ls[r] = 0.0; so_rs[r] = 0.0; so_ds[r] = so_dls[r] = so_pa[r] = 0.0; }
#define IS_SOFT(r) ((r) < NSOFT && so_Zw[(r) < NSOFT ? (r) : 0] >= 0.0)
// The first NSOFT slots of a lane are ONE-SIDED by construction of the table (api.hip: rebuild_slots puts a lane's one-sided slots -- the soft
// ones first -- there and nothing else): they live in the lower side's registers alone, an upper side as the lower side of the negated row
// (s_sg = -1: -R z >= -ub), and the upper side's seven registers per slot are never touched.  (The soft instantiations are register-bound:
// 648 B of scratch per lane in `<8,3,1,1>` before this.)
// (not in the PATH == 2 instantiation: `k_qp_wave<10,4,2,1>` in this form returned wrong statuses when built with LLVM's iterative ILP
// scheduler -- run to run different ones -- while the default build matched the CPU restatement on 4096 instances; the cause was not found
// (NOTES.md R4.10), so that kernel keeps both sides' registers for every slot, the form that passes in both builds)
#define ONE_SIDED(r) (NSOFT > 0 && PATH != 2 && (r) < NSOFT)
#define HAS_L(r) (ONE_SIDED(r) ? true : fin(s_dl[r]))
#define HAS_U(r) (ONE_SIDED(r) ? false : fin(s_du[r]))
#define ROW_DOT(r, v) (ONE_SIDED(r) ? s_sg[(r) < NSOFT ? (r) : 0] * row_dot(s_kc[r], (v)) : row_dot(s_kc[r], (v)))
#define ROW_SIGN(r, c) (ONE_SIDED(r) ? s_sg[(r) < NSOFT ? (r) : 0] * (c) : (c))
    double s_sg[NSO];
#pragma unroll
    for (int r = 0; r < NSO; r++) s_sg[r] = 1.0;
// rows are split (two slots, one lane, one LDS word) only when soft sides exist
#define SLOT_ACC(dst, v) do { if (NSOFT > 0) (dst) += (v); else (dst) = (v); } while (0)
    double r_ineq = 0.0, r_comp = 0.0;
#pragma unroll
    for (int r = 0; r < NSLOT; r++) {
        const int s = tid + NT * r;
        s_kc[r] = -1; s_dl[r] = -INFINITY; s_du[r] = INFINITY;
        lam_l[r] = lam_u[r] = 0.0; t_l[r] = t_u[r] = 1.0;
        const int kc = (s < a.nslots) ? a.slot_kc[s] : -1;
        if (kc >= 0) {
            const int k = kc >> 4, c = kc & 15;
            s_kc[r] = k * NCK + c;
            double cz;
            if (c < 8) cz = xb[k * 8 + c];
            else if (c < 10) cz = ub[k * 2 + c - 8];
            else if (ALAT && c == 14) {
                double g4[4];
                cz = alat_eval(xb[k * 8 + 3], xb[k * 8 + 4], xb[k * 8 + 6], xb[k * 8 + 7], g4);
            } else if (c >= 12) {
                const double n = xb[k * 8 + 1], psi = xb[k * 8 + 2];
                const double foot = -0.5 * a.car_L * sin(fabs(psi)), lat = 0.5 * a.car_W * cos(psi);
                cz = (c == 12) ? n + foot + lat - w_R : -n - foot + lat - w_L;
            } else {
                cz = 0.0;
#pragma unroll
                for (int j = 0; j < 8; j++) cz = fma(CDV(k, c - 10, j), xb[k * 8 + j], cz);
                cz = fma(CDV(k, c - 10, 8), ub[k * 2 + 0], cz);
                cz = fma(CDV(k, c - 10, 9), ub[k * 2 + 1], cz);
            }
            const double lb = a.slot_lb[s], ubd = a.slot_ub[s];
            bool soft = false;
            if (r < NSOFT) { so_zw[r < NSOFT ? r : 0] = a.slot_zw[s]; so_Zw[r < NSOFT ? r : 0] = a.slot_Zw[s]; soft = a.slot_Zw[s] >= 0.0; }
            // soft sides may be violated: they do not count as infeasibility of the iterate
            // (the incoming multipliers are fetched where they are used: as values of their own they were two loads per slot in every QP and moved the
            // all-hard kernels' register allocation -- slot bounds from the accumulator file to scratch, -1.6 % on the headline)
#define LAM_IN(up) ((ALAT && c == 14) ? lamab[k * 2 + (up)] : lamb[k * 28 + 14 * (up) + c])
            if (fin(lb)) { s_dl[r] = lb - cz; if (want_res && !soft) { r_ineq = fmax(r_ineq, s_dl[r]); r_comp = fmax(r_comp, fabs(LAM_IN(0) * s_dl[r])); } }
            if (fin(ubd)) { s_du[r] = ubd - cz; if (want_res && !soft) { r_ineq = fmax(r_ineq, -s_du[r]); r_comp = fmax(r_comp, fabs(LAM_IN(1) * s_du[r])); } }
#undef LAM_IN
            if (ONE_SIDED(r) && fin(ubd)) { s_sg[r < NSOFT ? r : 0] = -1.0; s_dl[r] = -s_du[r]; }      // R z <= ub  as  -R z >= -ub
        }
    }
    if (want_res) { r_stat = blk_max(r_stat); r_eq = blk_max(r_eq); r_ineq = blk_max(r_ineq); r_comp = blk_max(r_comp); }
    if (want_res && tid == 0) {
        double *rs = a.res + (size_t)b * 4;
        rs[0] = r_stat; rs[1] = r_eq; rs[2] = r_ineq; rs[3] = r_comp;
    }

    const double tol_g = a.tol * sg, tol_b = a.tol * sb, tol_d = a.tol * sb, tol_m = a.tol * sg;
    const double mu_floor = 0.1 * tol_m;
    const double mu0 = a.mu0 * sg;
    const double inv_m = (a.m_act > 0) ? 1.0 / a.m_act : 0.0;

    // R_c . v of the slot (k, c) for a stage-major vector v[NS][10] in LDS
    auto row_dot = [&](int kc, const double *v) -> double {
        const int k = kc / NCK, c = kc % NCK;
        if (c < 10) return v[k * 10 + c];
        if (ALAT && c == 14) return ha[k * 4] * v[k * 10 + 3] + ha[k * 4 + 1] * v[k * 10 + 4] + ha[k * 4 + 2] * v[k * 10 + 6] + ha[k * 4 + 3] * v[k * 10 + 7];
        if (PATH && c >= 12) return (c == 12) ? v[k * 10 + 1] + hc[k * 2] * v[k * 10 + 2] : -v[k * 10 + 1] + hc[k * 2 + 1] * v[k * 10 + 2];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 10; j++) acc = fma(CDV(k, c - 10, j), v[k * 10 + j], acc);
        return acc;
    };

/*@S:0*/
    // ------------------------------------------------------------------ initial point
    for (int e = tid; e < NS * 10; e += NT) z[e] = (e < 8) ? a.x0[(size_t)b * 8 + e] - xb[e] : 0.0;
    for (int e = tid; e < NS * 8; e += NT) pi[e] = 0.0;
    BSYNC();
#pragma unroll
    for (int r = 0; r < NSLOT; r++) {
        if (s_kc[r] < 0) continue;
        const double rz = ROW_DOT(r, z);
        const bool al = HAS_L(r), au = HAS_U(r);
        double tau_c = a.tau0;
        if (al && au) tau_c = fmin(a.tau0, 0.25 * (s_du[r] - s_dl[r]));
        if (IS_SOFT(r)) {
            const double slack = al ? rz - s_dl[r] : s_du[r] - rz;
            const double s0 = fmax(a.tau0, a.tau0 - slack);
            so_s[r < NSOFT ? r : 0] = s0;
            so_ls[r < NSOFT ? r : 0] = mu0 / s0;
            if (al) { t_l[r] = slack + s0; lam_l[r] = mu0 / t_l[r]; } else { t_u[r] = slack + s0; lam_u[r] = mu0 / t_u[r]; }
            continue;
        }
        if (al) { t_l[r] = fmax(rz - s_dl[r], tau_c); lam_l[r] = mu0 / t_l[r]; }
        if (au) { t_u[r] = fmax(s_du[r] - rz, tau_c); lam_u[r] = mu0 / t_u[r]; }
    }

    // ------------------------------------------------------------------ interior-point iterations
    int qstatus = 1, it = 0;
    bool exact_mode = false;     // residuals from the problem data in every iteration (set when followed residuals failed their check)
    double res_g = 0, res_b = 0, res_d = 0, res_m = 0, mu = 0;
    double fol_g = 0.0, fol_b = 0.0;        // this lane's share of max |r_g|, max |r_b| of the followed residuals (formed where the update writes them)
    double rd_l[NSLOT], rd_u[NSLOT], dlam_l[NSLOT], dlam_u[NSLOT], dt_l[NSLOT], dt_u[NSLOT];
    double pa_l[NSLOT], pa_u[NSLOT];      // dlam * dt of the predictor (only the products enter the corrector): 2 registers per slot less than the factors
    for (it = 0;; it++) {
/*@S:1*/
        // ---- slack residuals, complementarity; lam_l - lam_u -> cf (read by the exact stationarity residual only) ----
        bool exact = (it == 0) || exact_mode, finished = false;
        auto multipliers_to_cf = [&]() {
            for (int e = tid; e < NS * NCK; e += NT) cf[e] = 0.0;
            BSYNC();
#pragma unroll
            for (int r = 0; r < NSLOT; r++) {
                if (s_kc[r] < 0) continue;
                const bool al = HAS_L(r), au = HAS_U(r);
                SLOT_ACC(cf[s_kc[r]], ROW_SIGN(r, (al ? lam_l[r] : 0.0) - (au ? lam_u[r] : 0.0)));     // the two halves of a split slot share a lane
            }
            BSYNC();
        };
        // (LEAN == 0, the SQP instantiations of the persistent loop: the form before -- the multipliers go to LDS in every iteration, inside the
        // loop below, and the norms are taken in a pass of their own; the leaner form costs THAT kernel 13 % in spilled registers)
        if (LEAN != 0) { if (exact) multipliers_to_cf(); }
        else {
            for (int e = tid; e < NS * NCK; e += NT) cf[e] = 0.0;
            BSYNC();
        }
        double mu_acc = 0.0, res_gs = 0.0;
        res_d = 0.0; res_m = 0.0;
#pragma unroll
        for (int r = 0; r < NSLOT; r++) {
            rd_l[r] = rd_u[r] = 0.0;
            if (s_kc[r] < 0) continue;
            const double rz = ROW_DOT(r, z);
            const bool al = HAS_L(r), au = HAS_U(r);
            const double sv = IS_SOFT(r) ? so_s[r < NSOFT ? r : 0] : 0.0;      // the slack enters its (single) side
            // (the complementarity products are summed as rounded products, in slot order: the sum is then the same number whichever
            // lanes hold the slots -- one wave per instance or four, SLOT_SUM below)
            double pl = 0.0, pu = 0.0;
            if (al) { rd_l[r] = rz - t_l[r] - s_dl[r] + sv; pl = __dmul_rn(lam_l[r], t_l[r]); res_m = nanmax(res_m, fabs(pl)); }
            if (au) { rd_u[r] = s_du[r] - rz - t_u[r] + sv; pu = __dmul_rn(lam_u[r], t_u[r]); res_m = nanmax(res_m, fabs(pu)); }
            mu_acc = __dadd_rn(mu_acc, __dadd_rn(pl, pu));
            res_d = nanmax(res_d, nanmax(fabs(rd_l[r]), fabs(rd_u[r])));
            if (IS_SOFT(r)) {
                const int q = r < NSOFT ? r : 0;
                so_rs[q] = so_Zw[q] * so_s[q] + so_zw[q] - (al ? lam_l[r] : lam_u[r]) - so_ls[q];
                res_gs = nanmax(res_gs, fabs(so_rs[q]));
                mu_acc += so_ls[q] * so_s[q]; res_m = nanmax(res_m, fabs(so_ls[q] * so_s[q]));
            }
            if (LEAN == 0) SLOT_ACC(cf[s_kc[r]], ROW_SIGN(r, (al ? lam_l[r] : 0.0) - (au ? lam_u[r] : 0.0)));
        }
        if constexpr (NW > 1) {
            // several waves per instance: the slots' terms go to LDS (gam: dead between the factor sweep and the next coefficient phase) and every wave
            // sums them in the order of the 64-lane table
#pragma unroll
            for (int r = 0; r < NSLOT; r++) {
                const int sidx = tid + NT * r;
                if (sidx >= a.nslots_can) continue;
                const bool on = s_kc[r] >= 0;
                gam[sidx] = __dadd_rn((on && fin(s_dl[r])) ? __dmul_rn(lam_l[r], t_l[r]) : 0.0, (on && fin(s_du[r])) ? __dmul_rn(lam_u[r], t_u[r]) : 0.0);
            }
            __syncthreads();
            mu_acc = 0.0;
            for (int sidx = lane; sidx < a.nslots_can; sidx += 64) mu_acc = __dadd_rn(mu_acc, gam[sidx]);
            __syncthreads();
        }
/*@S:14*/
        if (LEAN == 0) BSYNC();
        // ---- stationarity and dynamics residuals ----
        // exact: formed from the problem data -- (i) the terms without [A B], g + H z - pi_k - R'(lam_l - lam_u), then (ii) [A B]'pi_{k+1}
        // and r_b from the records.  Otherwise they are what the update at the end of the previous iteration left in rgb / rb: the
        // Newton step solves the linearised rows exactly, so the residuals follow the step (no pass over the records).  Followed residuals
        // that pass the convergence test are formed from the data and tested again; should that fail, every later iteration forms them
        // from the data (exact_mode).
        for (int rpass = 0; rpass < 2; rpass++) {
            if (exact) {
                // (an unrolled variant that forms all values before the first store -- QP gradient entries and LDS operands of all passes in
                // flight together -- was 30 % faster here but cost the slot phases twice that in spilled slot registers)
                for (int e = tid; e < NS * 10; e += NT) {
                    const int k = e / 10, j = e % 10;
                    double acc = gb[e];
                    if (HL && h_sparse) {
                        const int row = ((k == N) ? 10 : 0) + j;
#pragma unroll
                        for (int q = 0; q < 3; q++) acc = fma(spv[row * 3 + q], z[k * 10 + spc[row * 3 + q]], acc);
                    } else {
#pragma unroll
                        for (int l = 0; l < 10; l++) acc = fma(HS(k, j, l), z[k * 10 + l], acc);
                    }
                    if (k < N) {
                        acc = fma(-CDV(k, 0, j), cf[k * NCK + 10], acc);
                        acc = fma(-CDV(k, 1, j), cf[k * NCK + 11], acc);
                    }
                    if (j < 8) acc -= pi[k * 8 + j];
                    acc -= cf[k * NCK + j];
                    if (PATH && (j == 1 || j == 2)) {
                        const double l12 = cf[k * NCK + 12], l13 = cf[k * NCK + 13];
                        acc -= (j == 1) ? l12 - l13 : hc[k * 2] * l12 + hc[k * 2 + 1] * l13;
                    }
                    if (ALAT && alat_slot(j) >= 0) acc -= ha[k * 4 + alat_slot(j)] * cf[k * NCK + 14];
                    gt[e] = acc;
                }
                BSYNC();
/*@S:2*/
                // (ii) [A B]' pi_{k+1} and the dynamics residual rb_k = A z_k + B u_k + b_k - z_{k+1} (LDS and slot 88 of the record, where
                // the factor sweep picks it up): one dot product per lane, all stages in parallel
                dyn_residual<4, NT>(N, tid, const_cast<double *>(linb), z, pi, gt, rb, LIN_REC);
                BSYNC();
            }       // (otherwise gt already holds the followed residual: the update at the end of the previous iteration left it there)
/*@S:3*/
            res_g = 0.0; res_b = 0.0;
            if (exact || LEAN == 0) {
                for (int e = tid; e < NS * 10; e += NT) {
                    const int k = e / 10, j = e % 10;
                    double v = gt[e];
                    if ((k == 0 && j < 8) || (k == N && j >= 8)) { v = 0.0; gt[e] = 0.0; }
                    if (exact) rgb[e] = v;          // the residual the update at the end of the iteration carries on
                    res_g = nanmax(res_g, fabs(v));
                }
                for (int e = tid; e < N * 8; e += NT) res_b = nanmax(res_b, fabs(rb[e]));
            } else { res_g = fol_g; res_b = fol_b; }        // the update that wrote the followed residuals took their norms (masked entries are stored as zeros)
            res_g = nanmax(res_g, res_gs);
            res_g = blk_nanmax(res_g); res_b = blk_nanmax(res_b);
            if (rpass == 0) { res_d = blk_nanmax(res_d); res_m = blk_nanmax(res_m); mu = ((NW == 1) ? blk_sum(mu_acc) : wave_sum(mu_acc)) * inv_m; }
            // not-a-number in the data of the QP (first pass over them): status 1; a QP that diverges on the way: a failed QP, status 4 --
            // the same two codes as the reference restatement, wherever the overflow first shows
            if (!(res_g == res_g) || !(res_b == res_b) || !(res_d == res_d) || !(res_m == res_m)) { qstatus = (it == 0) ? 3 : 4; finished = true; break; }
            const bool conv = res_g <= tol_g && res_b <= tol_b && res_d <= tol_d && res_m <= tol_m;
            if (conv && exact) { qstatus = 0; finished = true; break; }
            // followed residuals that say "converged" are formed from the data and tested again -- and so are those the iteration limit stops
            // at: what decides the loose acceptance below, and what ihm2mpc_get_qp_residuals reports, has been checked against A, B and R
            if ((conv || it >= a.iter_max) && !exact) {
                exact = true; exact_mode = true;
                if (LEAN != 0) multipliers_to_cf();
                continue;
            }
            break;
        }
        if (finished) break;
        if (it >= a.iter_max) { qstatus = 1; break; }
        BSYNC();
        // separate step lengths for the primal (z, t, s) and the dual (pi, lam, lam_s) variables, as HPIPM's split_step
        double alpha = 1.0, alpha_d = 1.0, sigma = 0.0;
/*@S:4*/
        // ---- barrier weights and gradient coefficients of the owned slots -> LDS (pass 0: predictor; pass 1: corrector) ----
        auto slot_coeffs = [&](int pass) {
            // ---- barrier weights and gradient coefficients of the owned slots -> LDS ----
            for (int e = tid; e < NS * NCK; e += NT) { cf[e] = 0.0; if (pass == 0) gam[e] = 0.0; }
            BSYNC();
            const double mu_t = fmax(sigma * mu, mu_floor);
#pragma unroll
            for (int r = 0; r < NSLOT; r++) {
                if (s_kc[r] < 0) continue;
                const bool al = HAS_L(r), au = HAS_U(r);
                double c = 0.0;
                if (IS_SOFT(r)) {
                    // eliminated slack block: gamma_eff = gam (Z + gam_s)/D, coef_eff = c1 - gam (rs + c1 + c2)/D
                    const int q = r < NSOFT ? r : 0;
                    const double lm = al ? lam_l[r] : lam_u[r], tt = al ? t_l[r] : t_u[r], rdv = al ? rd_l[r] : rd_u[r];
                    const double gm = lm / tt, gs = so_ls[q] / so_s[q], D = so_Zw[q] + gm + gs;
                    double c1, c2, rsv;
                    if (pass == 0) { c1 = (lm * tt + lm * rdv) / tt; c2 = so_ls[q]; rsv = so_rs[q]; gam[s_kc[r]] += gm * (so_Zw[q] + gs) / D; }
                    else { c1 = ((al ? pa_l[r] : pa_u[r]) - mu_t) / tt; c2 = (so_pa[q] - mu_t) / so_s[q]; rsv = 0.0; }
                    c = c1 - gm * (rsv + c1 + c2) / D;
                    cf[s_kc[r]] += ONE_SIDED(r) ? ROW_SIGN(r, c) : (al ? c : -c);
                    continue;
                }
                if (pass == 0) {
                    // one reciprocal per side serves the coefficient and the barrier weight
                    const double gl = al ? lam_l[r] / t_l[r] : 0.0, gu = au ? lam_u[r] / t_u[r] : 0.0;
                    if (al) c += fma(gl, rd_l[r], lam_l[r]);        // (lam t + lam rd) / t
                    if (au) c -= fma(gu, rd_u[r], lam_u[r]);
                    SLOT_ACC(gam[s_kc[r]], gl + gu);
                } else {
                    if (al) c += (pa_l[r] - mu_t) / t_l[r];
                    if (au) c -= (pa_u[r] - mu_t) / t_u[r];
                }
                SLOT_ACC(cf[s_kc[r]], ROW_SIGN(r, c));
            }
            BSYNC();
        };
        // modified gradient: gt += R' cf (pass 1 adds its increment on top of the predictor's gradient)
        auto add_coeffs = [&]() {
            for (int e = tid; e < NS * 10; e += NT) {
                const int k = e / 10, j = e % 10;
                double acc = gt[e] + cf[k * NCK + j];
                if (k < N) {
                    acc = fma(CDV(k, 0, j), cf[k * NCK + 10], acc);
                    acc = fma(CDV(k, 1, j), cf[k * NCK + 11], acc);
                }
                if (PATH && (j == 1 || j == 2)) {
                    const double c12 = cf[k * NCK + 12], c13 = cf[k * NCK + 13];
                    acc += (j == 1) ? c12 - c13 : hc[k * 2] * c12 + hc[k * 2 + 1] * c13;
                }
                if (ALAT && alat_slot(j) >= 0) acc += ha[k * 4 + alat_slot(j)] * cf[k * NCK + 14];
                gt[e] = acc;        // pass 1 adds its increment on top of the predictor's gradient
            }
            BSYNC();

        };
        slot_coeffs(0);
        add_coeffs();
/*@S:5*/
        // ---- factorisation and the predictor's vector recursion: P_k, M_k -> HBM/L2; K_k, Guu^-1, p_k, kff_k, c_k = rb_k - B kff_k
        // and P_{k+1} rb_k -> LDS (riccati_mfma.hpp).  The records' rb slots were written by this wave: wait for them. ----
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
        if (wv == 0) {
            // offsets of the carve-up above as plain integers (a difference of two generic pointers into LDS makes the compiler
            // build both flat addresses -- and mis-fold their null checks in the register-starved instantiations)
            RicLds L;
            L.gt = NS * 10; L.pv = NS * 28; L.gam = NS * 36 + N * 8; L.dz = L.gam + 2 * NS * NCK; L.kff = L.dz + NS * 10; L.Kl = L.kff + N * 4;
            L.Ginv = L.Kl + N * 16; L.hv = L.Ginv + N * 8; L.tile = L.hv + N * 8; L.hc = L.tile + 136; L.ha = L.hc + NS * 2;
            riccati_sweep_mfma<NCK, PATH != 0, UNI != 0, RIC_RING, ALAT>(N, lane, linb, a.Hs, a.CD, L, Pg, Mg, LIN_REC, a.m_act == 0, a.symmetrize != 0);
            if (lane < 8) dz[lane] = 0.0;
        }
        BSYNC();
        for (int pass = 0; pass < 2; pass++) {
            const double mu_t = fmax(sigma * mu, mu_floor);
            if (pass == 1) slot_coeffs(1);
            if (pass == 1) {
            add_coeffs();
/*@S:6*/
            // ---- corrector's vector recursion: p_k = gt_x - K'gt_u + M_k'(P_{k+1} rb_k + p_{k+1}) (the predictor's came out of
            // the factor sweep) ----
            // the part without p_{k+1} for all stages in parallel ...
            for (int e = tid; e < NS * 8; e += NT) {
                const int k = e >> 3, j = e & 7;
                double v = gt[k * 10 + j];
                if (k < N) v -= Kl[k * 16 + j] * gt[k * 10 + 8] + Kl[k * 16 + 8 + j] * gt[k * 10 + 9];
                pv[e] = v;
            }
            BSYNC();
/*@S:7*/
            // ... then the recursion itself, entirely in registers.  A lane is (g, w) = (lane >> 3, lane & 7).  The contraction runs
            // over w inside the group on even steps (result: one value per group g) and over g across the groups on odd steps
            // (result: one value per position w), so the output of a step is already laid out as the input of the next:
            //   even: lane holds M[w][g], q_w = Prb_k[w] + p_{k+1}[w] -> p_k[g] ;  odd: lane holds M[g][w], q_g -> p_k[w]
            if constexpr (LEAN != 0) {
            if (wv == 0) {
                const int g = lane >> 3, w = lane & 7;
                double pw = pv[N * 8 + w], pg = 0.0;
                // the stage's base term is added by ONE lane of each sum: the other lanes read a word that is zero throughout (dz[0]: dx_0 = 0)
                // instead of selecting -- and every lane of a sum stores its (identical) result: no select, no exec mask per stage
                // (running pointers with a per-lane decrement -- 16 doubles or none per two stages: the fetches come in the order k = N-1, N-2, ...
                // with even and odd steps alternating)
                const int se2 = (w == 0) ? 16 : 0, so2 = (g == 0) ? 16 : 0;
                const double *be = (w == 0) ? pv + (N - 1) * 8 + g : dz, *bo = (g == 0) ? pv + (N - 2) * 8 + w : dz;
                const double *qe = Prb + (N - 1) * 8 + w, *qo = Prb + (N - 2) * 8 + g;      // P_{k+1} rb_k, same order
                double *oe = pv + (N - 1) * 8 + g, *oo = pv + (N - 2) * 8 + w;               // results: k = N-1, N-3, ... and N-2, N-4, ...
                stream_rows<-1, SWEEP_RING, SWEEP_DL>(Mg, N, RIC_IDX(w, g), RIC_IDX(g, w),
                    [&](int k, bool odd, double &prb, double &base) {
                        if (odd) { prb = *qo; qo -= 16; base = *bo; bo -= so2; } else { prb = *qe; qe -= 16; base = *be; be -= se2; }
                    },
                    [&](int k, double m, bool odd, double prb, double base) {
                        // only the product with the carried value sits on the dependent chain: m * prb and the stage's base term are
                        // formed as soon as the operands arrive
                        if (!odd) {
                            pg = sum8(fma(m, pw, fma(m, prb, base)));
                            *oe = pg; oe -= 16;
                        } else {
                            pw = sum_stride8(fma(m, pg, fma(m, prb, base)));
                            *oo = pw; oo -= 16;
                        }
                    });
            }
            } else {
            if (wv == 0) {
                const int g = lane >> 3, w = lane & 7;
                double pw = pv[N * 8 + w], pg = 0.0;
                stream_rows_v1<-1, 8, 4>(Mg, N, RIC_IDX(w, g), RIC_IDX(g, w),
                    [&](int k, bool odd, double &prb, double &base) { prb = Prb[k * 8 + (odd ? g : w)]; base = pv[k * 8 + (odd ? w : g)]; },
                    [&](int k, double m, bool odd, double prb, double base) {
                        // only the product with the carried value sits on the dependent chain: m * prb and the stage's base term (added
                        // by ONE lane of each sum) are formed as soon as the operands arrive
                        if (!odd) {
                            const double off = fma(m, prb, (w == 0) ? base : 0.0);
                            pg = sum8(fma(m, pw, off));
                            if (w == 0) pv[k * 8 + g] = pg;
                        } else {
                            const double off = fma(m, prb, (g == 0) ? base : 0.0);
                            pw = sum_stride8(fma(m, pg, off));
                            if (g == 0) pv[k * 8 + w] = pw;
                        }
                    });
            }
            }
            BSYNC();
/*@S:8*/
            // feed-forward terms kff_k = Guu^-1 (gt_u + B'(P_{k+1} rb_k + p_{k+1})) and the affine part
            // c_k = rb_k - B kff_k of the forward recursion, all stages in parallel
            for (int k = tid; k < N; k += NT) {
                const double *rec = linb + (size_t)k * LIN_REC;
                double g0 = gt[k * 10 + 8], g1 = gt[k * 10 + 9];
                double Bk[16];
#pragma unroll
                for (int l = 0; l < 16; l++) Bk[l] = rec[64 + l];
#pragma unroll
                for (int l = 0; l < 8; l++) {
                    const double h = Prb[k * 8 + l] + pv[(k + 1) * 8 + l];
                    g0 = fma(Bk[l * 2 + 0], h, g0);
                    g1 = fma(Bk[l * 2 + 1], h, g1);
                }
                const double kf0 = Ginv[k * 8 + 0] * g0 + Ginv[k * 8 + 1] * g1;
                const double kf1 = Ginv[k * 8 + 1] * g0 + Ginv[k * 8 + 2] * g1;
                kff[k * 4 + 0] = kf0;
                kff[k * 4 + 1] = kf1;
#pragma unroll
                for (int i = 0; i < 8; i++) dz[(k + 1) * 10 + i] = rb[k * 8 + i] - Bk[i * 2] * kf0 - Bk[i * 2 + 1] * kf1;
            }
            if (tid < 8) dz[tid] = 0.0;
            BSYNC();
            }

/*@S:9*/
            // ---- forward recursion: dx_{k+1} = c_k + M_k dx_k, same alternating register layout ----
            //   even: lane holds M[g][w], dx_k[w] -> dx_{k+1}[g] ;  odd: lane holds M[w][g], dx_k[g] -> dx_{k+1}[w]
            if constexpr (LEAN != 0) {
            if (wv == 0) {
                const int g = lane >> 3, w = lane & 7;
                double dxw = dz[w], dxg = 0.0;
                // the affine term rides in ONE lane's product (an fma off the dependent chain's critical add); the other lanes read the zero word
                // dz[0] (dx_0 = 0), and every lane of a sum stores its (identical) result
                // (running pointers with a per-lane increment: the fetches come in the order k = 0, 1, ... with even and odd steps alternating)
                const int se2 = (w == 0) ? 20 : 0, so2 = (g == 0) ? 20 : 0;
                const double *ce = (w == 0) ? dz + 10 + g : dz, *co = (g == 0) ? dz + 20 + w : dz;
                stream_rows<+1, SWEEP_RING, SWEEP_DL>(Mg, N, RIC_IDX(g, w), RIC_IDX(w, g),
                    [&](int k, bool odd, double &c, double &unused) {
                        if (odd) { c = *co; co += so2; } else { c = *ce; ce += se2; }
                        unused = 0.0;
                    },
                    [&](int k, double m, bool odd, double c, double) {
                        if (!odd) {
                            dxg = sum8(fma(m, dxw, c));
                            dz[(k + 1) * 10 + g] = dxg;
                        } else {
                            dxw = sum_stride8(fma(m, dxg, c));
                            dz[(k + 1) * 10 + w] = dxw;
                        }
                    });
            }
            } else {
            if (wv == 0) {
                const int g = lane >> 3, w = lane & 7;
                double dxw = dz[w], dxg = 0.0;
                stream_rows_v1<+1, 8, 4>(Mg, N, RIC_IDX(g, w), RIC_IDX(w, g),
                    [&](int k, bool odd, double &c, double &unused) { c = dz[(k + 1) * 10 + (odd ? w : g)]; unused = 0.0; },
                    [&](int k, double m, bool odd, double c, double) {
                        // the affine term rides in ONE lane's product (an fma off the dependent chain's critical add)
                        if (!odd) {
                            dxg = sum8(fma(m, dxw, (w == 0) ? c : 0.0));
                            if (w == 0) dz[(k + 1) * 10 + g] = dxg;
                        } else {
                            dxw = sum_stride8(fma(m, dxg, (g == 0) ? c : 0.0));
                            if (g == 0) dz[(k + 1) * 10 + w] = dxw;
                        }
                    });
            }
            }
            BSYNC();
/*@S:10*/
            // inputs and costate steps of all stages in parallel
            const bool want_dpi = (pass == 1) || (a.m_act == 0);
            for (int e = tid; e < NS * 2; e += NT) {
                const int k = e >> 1, aa = e & 1;
                double acc = 0.0;
                if (k < N) {
                    acc = -kff[k * 4 + aa];
#pragma unroll
                    for (int l = 0; l < 8; l++) acc = fma(-Kl[k * 16 + aa * 8 + l], dz[k * 10 + l], acc);
                }
                dz[k * 10 + 8 + aa] = acc;
            }
/*@S:16*/
            if (want_dpi) {
                // dpi_k = P_k dx_k + p_k.  P_k is symmetric: column i of its RIC_IDX layout holds the pairs (P[l][i], P[l+4][i])
                // adjacent -- four 16-byte loads per entry, those of three entries in flight before the first is used
                const int n = NS * 8;
                for (int base = 0; base < n; base += 3 * NT) {
                    double2 pr[3][4];
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        const int e = min(base + NT * q + tid, n - 1), k = e >> 3, i = e & 7;
#pragma unroll
                        for (int l = 0; l < 4; l++) pr[q][l] = *reinterpret_cast<const double2 *>(Pg + (size_t)k * 64 + RIC_IDX(l, i));
                    }
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        const int e = base + NT * q + tid, ec = min(e, n - 1), k = ec >> 3;
                        double acc = pv[ec];
#pragma unroll
                        for (int l = 0; l < 4; l++) { acc = fma(pr[q][l].x, dz[k * 10 + l], acc); acc = fma(pr[q][l].y, dz[k * 10 + l + 4], acc); }
                        if (e < n) pv[e] = acc;        // dpi_k = P_k dx_k + p_k
                    }
                }
            }
            BSYNC();

/*@S:11*/
            // ---- slack / multiplier steps, step length ----
            double amax = 1.0, amax_d = 1.0, mu_aff = 0.0, rmax = 1.0, rmax_d = 1.0;
#pragma unroll
            for (int r = 0; r < NSLOT; r++) {
                dlam_l[r] = dlam_u[r] = dt_l[r] = dt_u[r] = 0.0;
                if (s_kc[r] < 0) continue;
                const double drz = ROW_DOT(r, dz);
                if (IS_SOFT(r)) {
                    const int q = r < NSOFT ? r : 0;
                    const bool al = HAS_L(r);
                    const double lm = al ? lam_l[r] : lam_u[r], tt = al ? t_l[r] : t_u[r], rdv = al ? rd_l[r] : rd_u[r];
                    const double y = al ? drz : -drz;
                    const double gm = lm / tt, gs = so_ls[q] / so_s[q], D = so_Zw[q] + gm + gs;
                    const double rm1 = (pass == 0) ? lm * tt : lm * tt + (al ? pa_l[r] : pa_u[r]) - mu_t;
                    const double rm2 = (pass == 0) ? so_ls[q] * so_s[q] : so_ls[q] * so_s[q] + so_pa[q] - mu_t;
                    const double c1 = (rm1 + lm * rdv) / tt, c2 = rm2 / so_s[q];
                    const double dsv = -(so_rs[q] + c1 + c2) / D - gm / D * y;
                    const double dlsv = -(rm2 + so_ls[q] * dsv) / so_s[q];
                    const double dtv = y + dsv + rdv;
                    const double dlv = -(rm1 + lm * dtv) / tt;
                    so_ds[q] = dsv; so_dls[q] = dlsv;
                    if (al) { dt_l[r] = dtv; dlam_l[r] = dlv; } else { dt_u[r] = dtv; dlam_u[r] = dlv; }
                    if (dsv < 0.0) amax = fmin(amax, -so_s[q] / dsv);
                    if (dlsv < 0.0) amax_d = fmin(amax_d, -so_ls[q] / dlsv);
                    if (dtv < 0.0) amax = fmin(amax, -tt / dtv);
                    if (dlv < 0.0) amax_d = fmin(amax_d, -lm / dlv);
                    continue;
                }
                if (HAS_L(r)) {
                    const double rm = (pass == 0) ? lam_l[r] * t_l[r] : lam_l[r] * t_l[r] + pa_l[r] - mu_t;
                    dt_l[r] = drz + rd_l[r];
                    if (NSOFT == 0) {       // all-hard tables: one reciprocal per side, step bound as 1 / max(-dt / t)
                        const double it = 1.0 / t_l[r];
                        dlam_l[r] = -(rm + lam_l[r] * dt_l[r]) * it;
                        rmax = fmax(rmax, -dt_l[r] * it);
                        rmax_d = fmax(rmax_d, -dlam_l[r] / lam_l[r]);
                    } else {                // (the soft instantiations are register-bound: two more live values cost more than they save)
                        dlam_l[r] = -(rm + lam_l[r] * dt_l[r]) / t_l[r];
                        if (dt_l[r] < 0.0) amax = fmin(amax, -t_l[r] / dt_l[r]);
                        if (dlam_l[r] < 0.0) amax_d = fmin(amax_d, -lam_l[r] / dlam_l[r]);
                    }
                }
                if (HAS_U(r)) {
                    const double rm = (pass == 0) ? lam_u[r] * t_u[r] : lam_u[r] * t_u[r] + pa_u[r] - mu_t;
                    dt_u[r] = -drz + rd_u[r];
                    if (NSOFT == 0) {
                        const double it = 1.0 / t_u[r];
                        dlam_u[r] = -(rm + lam_u[r] * dt_u[r]) * it;
                        rmax = fmax(rmax, -dt_u[r] * it);
                        rmax_d = fmax(rmax_d, -dlam_u[r] / lam_u[r]);
                    } else {
                        dlam_u[r] = -(rm + lam_u[r] * dt_u[r]) / t_u[r];
                        if (dt_u[r] < 0.0) amax = fmin(amax, -t_u[r] / dt_u[r]);
                        if (dlam_u[r] < 0.0) amax_d = fmin(amax_d, -lam_u[r] / dlam_u[r]);
                    }
                }
            }
/*@S:15*/
            // hard sides collect max(-dt/t), max(-dlam/lam) (>= 1 matters only); soft sides the step bounds themselves
            if (NSOFT == 0) { amax = fmin(amax, 1.0 / rmax); amax_d = fmin(amax_d, 1.0 / rmax_d); }
            amax = blk_min(amax); amax_d = blk_min(amax_d);
            if (pass == 0) {
                if (a.m_act == 0) { alpha = alpha_d = 1.0; break; }
#pragma unroll
                for (int r = 0; r < NSLOT; r++) {
                    pa_l[r] = dlam_l[r] * dt_l[r]; pa_u[r] = dlam_u[r] * dt_u[r];
                    if (s_kc[r] < 0) continue;
                    mu_aff = __dadd_rn(mu_aff, __dadd_rn(HAS_L(r) ? __dmul_rn(fma(amax_d, dlam_l[r], lam_l[r]), fma(amax, dt_l[r], t_l[r])) : 0.0,
                                                         HAS_U(r) ? __dmul_rn(fma(amax_d, dlam_u[r], lam_u[r]), fma(amax, dt_u[r], t_u[r])) : 0.0));
                    if (IS_SOFT(r)) {
                        const int q = r < NSOFT ? r : 0;
                        so_pa[q] = so_dls[q] * so_ds[q];
                        mu_aff += (so_ls[q] + amax_d * so_dls[q]) * (so_s[q] + amax * so_ds[q]);
                    }
                }
                if constexpr (NW > 1) {        // the same sum in the order of the 64-lane table
#pragma unroll
                    for (int r = 0; r < NSLOT; r++) {
                        const int sidx = tid + NT * r;
                        if (sidx >= a.nslots_can) continue;
                        const bool on = s_kc[r] >= 0;
                        gam[sidx] = __dadd_rn((on && fin(s_dl[r])) ? __dmul_rn(fma(amax_d, dlam_l[r], lam_l[r]), fma(amax, dt_l[r], t_l[r])) : 0.0,
                                              (on && fin(s_du[r])) ? __dmul_rn(fma(amax_d, dlam_u[r], lam_u[r]), fma(amax, dt_u[r], t_u[r])) : 0.0);
                    }
                    __syncthreads();
                    mu_aff = 0.0;
                    for (int sidx = lane; sidx < a.nslots_can; sidx += 64) mu_aff = __dadd_rn(mu_aff, gam[sidx]);
                    __syncthreads();
                    mu_aff = wave_sum(mu_aff) * inv_m;
                } else
                mu_aff = blk_sum(mu_aff) * inv_m;
                const double ratio = (mu > 0.0) ? mu_aff / mu : 0.0;
                sigma = ratio * ratio * ratio;
            } else {
                alpha = fmin(1.0, IHM2MPC_IPM_STEP_FRACTION * amax); alpha_d = fmin(1.0, IHM2MPC_IPM_STEP_FRACTION * amax_d);
            }
            BSYNC();
        }
/*@S:12*/
        if (fmin(alpha, alpha_d) < 1e-12) { qstatus = 2; break; }
/*@S:17*/
        if (!exact_mode) {
            // residuals of the new iterate: r_g <- (1 - alpha_d) r_g + (alpha - alpha_d) H dz -> rgb (HBM/L2: the modified gradient in gt is
            // dead by now, but r_g itself has to survive the next iteration's gradient modifications) and gt; r_b <- (1 - alpha) r_b -> LDS and
            // the records' slot.  The old values of a batch are loaded before the first store (the stores would pin every later load behind
            // them).  (Loading all of them ahead of the iterate's own update, to cover their latency, was tried: the eight live values spill
            // the slot registers -- 904 k against 965 k solves/s over 20 steps.)
            const double cd = 1.0 - alpha_d, cp = alpha - alpha_d, cb = 1.0 - alpha;
            const int n10 = NS * 10;
            fol_g = 0.0; fol_b = 0.0;
            for (int base = 0; base < n10; base += 4 * NT) {
                double ro[4];
#pragma unroll
                for (int q = 0; q < 4; q++) ro[q] = rgb[min(base + NT * q + tid, n10 - 1)];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int e = base + NT * q + tid, ec = min(e, n10 - 1), k = ec / 10, j = ec % 10;
                    double hdz = 0.0;
                    if (HL && h_sparse) {
                        const int row = ((k == N) ? 10 : 0) + j;
#pragma unroll
                        for (int l = 0; l < 3; l++) hdz = fma(spv[row * 3 + l], dz[k * 10 + spc[row * 3 + l]], hdz);
                    } else {
#pragma unroll
                        for (int l = 0; l < 10; l++) hdz = fma(HS(k, j, l), dz[k * 10 + l], hdz);
                    }
                    const bool masked = (k == 0 && j < 8) || (k == N && j >= 8);
                    const double v = masked ? 0.0 : fma(cp, hdz, cd * ro[q]);
                    if (e < n10) { rgb[e] = v; gt[e] = v; if (LEAN != 0) fol_g = nanmax(fol_g, fabs(v)); }
                }
            }
            for (int e = tid; e < N * 8; e += NT) {
                const double v = cb * rb[e];
                rb[e] = v;
                if (LEAN != 0) fol_b = nanmax(fol_b, fabs(v));
                const_cast<double *>(linb)[(size_t)(e >> 3) * LIN_REC + RIC_REC_RB + (e & 7)] = v;
            }
        }
        for (int e = tid; e < NS * 10; e += NT) z[e] = fma(alpha, dz[e], z[e]);
        for (int e = tid; e < NS * 8; e += NT) pi[e] = fma(alpha_d, pv[e], pi[e]);
#pragma unroll
        for (int r = 0; r < NSLOT; r++) {
            if (s_kc[r] < 0) continue;
            if (HAS_L(r)) { lam_l[r] = fma(alpha_d, dlam_l[r], lam_l[r]); t_l[r] = fma(alpha, dt_l[r], t_l[r]); }
            if (HAS_U(r)) { lam_u[r] = fma(alpha_d, dlam_u[r], lam_u[r]); t_u[r] = fma(alpha, dt_u[r], t_u[r]); }
            if (IS_SOFT(r)) { const int q = r < NSOFT ? r : 0; so_s[q] = fma(alpha, so_ds[q], so_s[q]); so_ls[q] = fma(alpha_d, so_dls[q], so_ls[q]); }
        }
        BSYNC();
    }
    if (qstatus == 1 && !(res_g <= 1e4 * tol_g && res_b <= 1e4 * tol_b && res_d <= 1e4 * tol_d && res_m <= 1e4 * tol_m)) qstatus = 4;

/*@S:13*/
    // ------------------------------------------------------------------ RTI update
    int st = 0;
    if (qstatus == 3) st = 1;
    else if (qstatus == 2 || qstatus == 4) st = 4;
    BSYNC();
    if (st == 0) {
        double bad = 0.0;
        for (int e = tid; e < NS * 10; e += NT) bad = fmax(bad, isfinite(z[e]) ? 0.0 : 1.0);
        if (blk_max(bad) > 0.0) st = 1;
    }
    double *xw = a.x + (size_t)b * NS * 8, *uw = a.u + (size_t)b * N * 2;
    if (NSOFT > 0) {        // slacks of a failed instance read 0; an all-hard table never touches the array
        for (int e = tid; e < NS * 28; e += NT) slkb[e] = 0.0;
        if (ALAT) for (int e = tid; e < NS * 2; e += NT) slkab[e] = 0.0;
    }
    if (st == 0) {
        for (int e = tid; e < NS * 10; e += NT) {
            const int k = e / 10, j = e % 10;
            if (j < 8) xw[k * 8 + j] += z[e];
            else if (k < N) uw[k * 2 + j - 8] += z[e];
        }
        for (int e = tid; e < NS * 8; e += NT) pib[e] = (e < 8) ? 0.0 : pi[e];
        for (int e = tid; e < NS * 28; e += NT) lamb[e] = 0.0;
        if (ALAT) for (int e = tid; e < NS * 2; e += NT) lamab[e] = 0.0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NSLOT; r++) {
            if (s_kc[r] < 0) continue;
            const int k = s_kc[r] / NCK, c = s_kc[r] % NCK;
            if (ONE_SIDED(r)) {        // the one side, kept in the lower side's registers: s_sg < 0 = it is the row's upper side
                const int up = (s_sg[r < NSOFT ? r : 0] < 0.0) ? 1 : 0;
                if (ALAT && c == 14) { lamab[k * 2 + up] = lam_l[r]; if (IS_SOFT(r)) slkab[k * 2 + up] = so_s[r < NSOFT ? r : 0]; }
                else { lamb[k * 28 + 14 * up + c] = lam_l[r]; if (IS_SOFT(r)) slkb[k * 28 + 14 * up + c] = so_s[r < NSOFT ? r : 0]; }
                continue;
            }
            if (ALAT && c == 14) {
                if (fin(s_dl[r])) lamab[k * 2] = lam_l[r];
                if (fin(s_du[r])) lamab[k * 2 + 1] = lam_u[r];
                if (IS_SOFT(r)) slkab[k * 2 + (fin(s_dl[r]) ? 0 : 1)] = so_s[r < NSOFT ? r : 0];
                continue;
            }
            if (fin(s_dl[r])) lamb[k * 28 + c] = lam_l[r];
            if (fin(s_du[r])) lamb[k * 28 + 14 + c] = lam_u[r];
            if (IS_SOFT(r)) slkb[k * 28 + (fin(s_dl[r]) ? c : 14 + c)] = so_s[r < NSOFT ? r : 0];
        }
    }
    __syncthreads();
    if (tid < 2) a.u0[(size_t)b * 2 + tid] = uw[tid];
    if (tid == 0) {
        a.status[b] = st; a.qp_iter[b] = it;
        // the QP's own KKT residuals where the iteration stopped, relative to the scales its tolerance is taken against
        // (stationarity, dynamics, inequalities, complementarity: <= ipm_tol each for status 0)
        double *q = a.qp_res + (size_t)b * 4;
        q[0] = res_g / tol_g * a.tol; q[1] = res_b / tol_b * a.tol; q[2] = res_d / tol_d * a.tol; q[3] = res_m / tol_m * a.tol;      // tol_x = tol * scale_x
    }
}

#undef BSYNC

template <int NSLOT, int NSOFT, int PATH, int UNI>
__global__ __launch_bounds__(64) void k_qp_wave(QpArgs a)
{
    extern __shared__ double sm[];
    if ((int)blockIdx.x >= a.B) return;
    qp_wave_body<NSLOT, NSOFT, PATH, UNI>(a, blockIdx.x, sm);
}

// The latency kernel: NW wavefronts per instance (qp_wave_body with NW > 1), for batches that leave most of the chip idle -- the
// reference's own use is ONE car at 20 Hz.  Each wave sits on its own SIMD of the CU and keeps the full register budget.
#if QP_SET == 0
template <int NSLOT, int UNI, int NW>
__global__ __launch_bounds__(64 * NW) void k_qp_block(QpArgs a)
{
    extern __shared__ double sm[];
    if ((int)blockIdx.x >= a.B) return;          // block-uniform
    qp_wave_body<NSLOT, 0, 0, UNI, NW>(a, blockIdx.x, sm);
}
#endif

// ---- persistent per-instance loop: n_steps control steps of the MiL loop (python/main.py:476-517) in ONE launch ----
// A wavefront owns an instance and runs, step after step,  [lap wrap] -> plant (lane 0) -> reference ramp + warm-start shift
// -> linearisation (lane k = interval k) -> QP -> history.  Nothing couples two instances, so nothing makes a wave wait for
// another one: with one launch per phase the whole batch waits, every step, for its slowest QP (17-18 interior-point iterations
// when the mean is 9); here the fast instances run ahead and the batch time follows the MEAN iteration count.  The phases are
// the device functions the stand-alone kernels call (device_steps.hpp, qp_wave_body): the results are those of n_steps calls
// of ihm2mpc_step.
// The integrator and the dynamic plants are CALLED, not inlined: each gets its own register allocation instead of sharing one
// with the QP body (inlined, the three together spilled 1.4 KB per lane into the QP's loops); a call per step costs nothing.
__device__ __noinline__ void call_integrate_fkin6(const double *xk, const double *uk, const double *x_next, int tid, int M, double dt, int nknots,
                                                  const double *s_ref, const double *kappa_ref, double *rec, double *xn_out)
{
    dev_integrate_sens<IHM2MPC_MODEL_FKIN6>(xk, uk, x_next, tid, M, dt, nknots, s_ref, kappa_ref, rec, xn_out, nullptr);
}
// The dynamic OCP models (python/models.py:455-606; fdyn6u = the named deviation): dev_integrate_sens with the sub-step's base sensitivities
// parked in LDS (Sl: the QP's LDS, idle during the linearisation; 55 words per lane) -- the arithmetic of k_linearize_dyn.
template <int MODEL>
__device__ __noinline__ void call_integrate_dyn(const double *xk, const double *uk, const double *x_next, int tid, int M, double dt, int nknots,
                                                const double *s_ref, const double *kappa_ref, double *rec, double *Sl)
{
    dev_integrate_sens<MODEL>(xk, uk, x_next, tid, M, dt, nknots, s_ref, kappa_ref, rec, nullptr, Sl);
}
// the collocation integrator in the loop: the wave's 16 quads take the intervals base .. base + 15 (kernels_irk.hip: four lanes per interval)
template <int MODEL>
__device__ __noinline__ void call_linearize_irk(const IrkTab *tab, int b, int base, int N, int nknots, const double *s_ref, const double *kappa_ref, int tid,
                                                const double *x, const double *u, double *lin)
{
    const int st = threadIdx.x & 3, q = base + ((int)threadIdx.x >> 2), k = min(q, N - 1);
    const IrkRows rows = irk_rows_from(tab, st);
    irk_linearize_quad<MODEL>(st, rows, x + ((size_t)b * (N + 1) + k) * 8, u + ((size_t)b * N + k) * 2, tid, nknots, s_ref, kappa_ref,
                              lin + ((size_t)b * N + k) * LIN_REC, q < N);
}
// (the lanes 0..3 of the wave: the dynamic plant's wheels are spread over a quad, device_steps.hpp)
__device__ __noinline__ void call_sim_step(int b, int q, int model, int M, double dt, int nknots, const double *s_ref, const double *kappa_ref,
                                           const int32_t *track_id, const double *xs, const double *us, double *xn)
{
    dev_sim_step(b, q, model, M, dt, nknots, s_ref, kappa_ref, track_id, xs, us, xn, nullptr);
}

// the plant by collocation: every quad of the wave integrates the instance's control period (same arithmetic in all sixteen: the DPP
// exchanges want whole quads), lane 0 stores the new state -- the body of k_sim_irk (kernels_irk.hip)
__device__ __noinline__ void call_sim_irk(const IrkTab *tab, int b, int model, int M, int nknots, const double *s_ref, const double *kappa_ref,
                                          const int32_t *track_id, const double *u0, double *x0)
{
    const int st = threadIdx.x & 3;
    double x[8];
#pragma unroll
    for (int a = 0; a < 8; a++) x[a] = x0[(size_t)b * 8 + a];
    const double u_T = u0[(size_t)b * 2], u_d = u0[(size_t)b * 2 + 1];
    const int tid = track_id[b];
    TrackSeg trk;
    trk.init(s_ref + (size_t)tid * nknots, kappa_ref + (size_t)tid * nknots, nknots, x[0]);
    const IrkRows rows = irk_rows_from(tab, st);
    irk_sim_quad(st, rows, model, M, x, u_T, u_d, trk);
    __syncthreads();            // every lane has read the old state
    if (threadIdx.x == 0)
#pragma unroll
        for (int a = 0; a < 8; a++) x0[(size_t)b * 8 + a] = x[a];
}

template <int MODEL, bool ROLL>
__device__ __noinline__ void call_line_search(const LsArgs &ls, int b, int it, int last)
{
    line_search_body<MODEL, ROLL>(ls, b, it, last);
}

struct StepArgs {
    int n_steps, model, M_sim, M, nknots, lap_wrap, freeze;
    int ocp_model;                          // the model of the shooting intervals (IHM2MPC_MODEL_FKIN6 / FDYN6 / FDYN6U); `model` is the plant's
    int sqp_iters;                          // 0: one RTI iteration per step; > 0: SQP mode, that many iterations with the line search
    double s_target, dt, lap_stop;
    const double *s_ref, *kappa_ref;
    double *x0, *yref, *yref_e, *lin;      // the same arrays as QpArgs', writable
    int32_t *active;                        // (B) or nullptr = all active
    double *hist_u0, *hist_x0;              // (n_steps,B,2), (n_steps,B,8) or nullptr
    int32_t *hist_st, *hist_it;             // (n_steps,B) or nullptr
    const IrkTab *irk_tab;                  // IRK = 1: the tableau of the shooting intervals' collocation step, in device memory
    const IrkTab *sim_irk_tab;              // plant steps by collocation (python/main.py:395-400: Radau IIA x M_sim) instead of RK4 x M_sim; nullptr: RK4
};

// SQP = 0: one RTI iteration per step (the SQP code is compiled out: next to the QP body it changed the register allocation of
// the whole kernel and tripled the step time); SQP = 1: sqp_iters iterations with the KKT test and the line search.
// IRK = 1: the shooting intervals are integrated by the collocation step of kernels_irk.hip (three passes of 16 quads); the kinematic plant
// stays RK4 x M_sim and takes a phase of its own on lane 0 like the dynamic plants (the state-only rollout: it no longer rides along
// with the interval lanes).
// DYN = 1: the shooting intervals carry a dynamic OCP model (StepArgs.ocp_model: fdyn6 or fdyn6u, chosen per launch); a template parameter so
// that the kinematic kernels stay what they were (the run-time choice alone cost the benchmarked kernel 2 %: 988 k -> 968 k solves/s).
template <int NSLOT, int NSOFT, int PATH, int UNI, int SQP, int IRK = 0, int DYN = 0>
__global__ __launch_bounds__(64) void k_steps(const StepArgs *sp, QpArgs a, const LsArgs *lsp)
{
    // the loop's own arguments are read from device memory where they are used: as by-value kernel arguments they stayed in
    // registers across the QP body (50 spill reloads inside its loops that k_qp_wave does not have)
    const StepArgs &s = *sp;
    const LsArgs &ls = *lsp;       // in device memory: a by-value kernel argument whose address is taken would be copied to scratch
    extern __shared__ double sm[];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= a.B) return;
    const int N = a.N;
    const size_t B = a.B;
    // a car that stops (freeze) keeps its state: the rest of the history repeats it with zero inputs
    auto stop_from = [&](int step) {
        if (s.active && lane == 0) s.active[b] = 0;
        for (int t = step; t < s.n_steps; t++) {
            if (s.hist_u0 && lane < 2) s.hist_u0[((size_t)t * B + b) * 2 + lane] = 0.0;
            if (s.hist_x0 && lane < 8) s.hist_x0[((size_t)t * B + b) * 8 + lane] = s.x0[(size_t)b * 8 + lane];
            if (s.hist_st && lane == 0) s.hist_st[(size_t)t * B + b] = a.status[b];
            if (s.hist_it && lane == 0) s.hist_it[(size_t)t * B + b] = 0;
        }
    };
    for (int step = 0; step < s.n_steps; step++) {
        // wave-uniform: is this car still driving?  (freeze: a failed solve stops it, python/main.py:326-328)
        bool act = s.active ? s.active[b] != 0 : true;
        if (s.freeze && act) {
            const int st = a.status[b];
            if (st != 0 && st != 2) act = false;
        }
        if (!act && s.freeze) { stop_from(step); return; }        // without freeze a masked car keeps solving, only its plant stands still (as ihm2mpc_step)
        if (s.lap_wrap) { dev_wrap_lap(b, lane, N, s.nknots, s.s_ref, a.track_id, s.x0, a.x); __syncthreads(); }
        const double x_old = (lane < 8) ? s.x0[(size_t)b * 8 + lane] : 0.0;
        // The kinematic plant (model 0) is one more "interval" of the linearisation -- lane N integrates (x0, u0) with the
        // code of the interval lanes, in lockstep with them -- so it costs no time; the dynamic plants take a phase of their own.
        const bool irk_plant = s.sim_irk_tab != nullptr;
        const bool kin_plant = !IRK && !irk_plant && s.model == IHM2MPC_MODEL_FKIN6;
        double *spare = s.lin + (size_t)B * N * LIN_REC;
        if (!kin_plant && act) {
            if (irk_plant) call_sim_irk(s.sim_irk_tab, b, s.model, s.M_sim, s.nknots, s.s_ref, s.kappa_ref, a.track_id, a.u0, s.x0);
            else if (lane < 4) call_sim_step(b, lane, s.model, s.M_sim, s.dt, s.nknots, s.s_ref, s.kappa_ref, a.track_id, s.x0, a.u0, s.x0);
            __syncthreads();
        }
        dev_prepare(b, lane, N, s.s_target, 2, s.x0, a.x, a.u, s.yref, s.yref_e);      // warm-start shift
        const int n_it = SQP ? s.sqp_iters : 1;
        if (SQP && lane == 0) { ls.done[b] = 0; ls.sqp_iter[b] = 0; ls.qp_acc[b] = 0; }      // per-solve bookkeeping of the SQP mode
        __syncthreads();
        for (int it = 0; it < n_it; it++) {
            if (SQP) {      // the iterate the QP is built at: the line search walks from it towards the QP's full step
                const int NS = N + 1;
                for (int e = lane; e < NS * 8; e += 64) {
                    ((double *)ls.xp)[(size_t)b * NS * 8 + e] = a.x[(size_t)b * NS * 8 + e];
                    ((double *)ls.pip)[(size_t)b * NS * 8 + e] = a.pi[(size_t)b * NS * 8 + e];
                }
                for (int e = lane; e < N * 2; e += 64) ((double *)ls.up)[(size_t)b * N * 2 + e] = a.u[(size_t)b * N * 2 + e];
                for (int e = lane; e < NS * 28; e += 64) {
                    ((double *)ls.lamp)[(size_t)b * NS * 28 + e] = a.lam[(size_t)b * NS * 28 + e];
                    ((double *)ls.slkp)[(size_t)b * NS * 28 + e] = a.slk[(size_t)b * NS * 28 + e];
                }
            }
            {
                const int tid = a.track_id[b];
                const bool with_plant = kin_plant && it == 0 && act;
                const int om = DYN ? s.ocp_model : IHM2MPC_MODEL_FKIN6;         // wave-uniform
                if (IRK) {
                    for (int base = 0; base < N; base += 16) {
                        if constexpr (!DYN) call_linearize_irk<IHM2MPC_MODEL_FKIN6>(s.irk_tab, b, base, N, s.nknots, s.s_ref, s.kappa_ref, tid, a.x, a.u, s.lin);
                        else {
                            if (om == IHM2MPC_MODEL_FDYN6U) call_linearize_irk<IHM2MPC_MODEL_FDYN6U>(s.irk_tab, b, base, N, s.nknots, s.s_ref, s.kappa_ref, tid, a.x, a.u, s.lin);
                            else call_linearize_irk<IHM2MPC_MODEL_FDYN6>(s.irk_tab, b, base, N, s.nknots, s.s_ref, s.kappa_ref, tid, a.x, a.u, s.lin);
                        }
                    }
                } else
                for (int k = lane; k < N + (with_plant ? 1 : 0); k += 64) {
                    const bool plant = k == N;
                    const double *xk = plant ? s.x0 + (size_t)b * 8 : a.x + ((size_t)b * (N + 1) + k) * 8;
                    const double *uk = plant ? a.u0 + (size_t)b * 2 : a.u + ((size_t)b * N + k) * 2;
                    double *rec = plant ? spare + (size_t)b * LIN_REC : s.lin + ((size_t)b * N + k) * LIN_REC;
                    // (the kinematic plant of a dynamic OCP is the fkin6 integrator on its own lane, after the interval lanes)
                    if (!DYN || plant)
                        call_integrate_fkin6(xk, uk, plant ? xk : xk + 8, tid, plant ? s.M_sim : s.M, s.dt, s.nknots, s.s_ref, s.kappa_ref, rec,
                                             plant ? s.x0 + (size_t)b * 8 : nullptr);
                    else if constexpr (DYN != 0) {
                        if (om == IHM2MPC_MODEL_FDYN6U)
                            call_integrate_dyn<IHM2MPC_MODEL_FDYN6U>(xk, uk, xk + 8, tid, s.M, s.dt, s.nknots, s.s_ref, s.kappa_ref, rec, sm + lane);
                        else
                            call_integrate_dyn<IHM2MPC_MODEL_FDYN6>(xk, uk, xk + 8, tid, s.M, s.dt, s.nknots, s.s_ref, s.kappa_ref, rec, sm + lane);
                    }
                }
            }
            __syncthreads();
            if (it == 0) {
                if (s.freeze) {     // python/main.py:503-504: a NaN plant state stops the car where it was
                    const double v = (lane < 8) ? s.x0[(size_t)b * 8 + lane] : 0.0;
                    if (__any(v != v)) {
                        if (lane < 8) s.x0[(size_t)b * 8 + lane] = x_old;
                        __syncthreads();
                        stop_from(step);
                        return;
                    }
                }
                dev_prepare(b, lane, N, s.s_target, 1, s.x0, a.x, a.u, s.yref, s.yref_e);      // reference ramp from the new x0
                __syncthreads();
            }
            // LEAN (the sweeps and norm phases on a diet): measured per class of instantiation -- it gains 6-8 % in the all-hard RTI loops and costs the
            // SQP loops 13-17 % and the soft / track-row loops 2-9 % (their register allocation tips into scratch); the stand-alone QP kernels take it
            qp_wave_body<NSLOT, NSOFT, PATH, UNI, 1, SQP ? 0 : 1>(a, b, sm, SQP || step + 1 == s.n_steps);
            __syncthreads();
            if (SQP) {
                const int last = it == n_it - 1;
                if constexpr (!DYN) call_line_search<IHM2MPC_MODEL_FKIN6, IRK != 0>(ls, b, it, last);
                else {
                    if (s.ocp_model == IHM2MPC_MODEL_FDYN6U) call_line_search<IHM2MPC_MODEL_FDYN6U, IRK != 0>(ls, b, it, last);
                    else call_line_search<IHM2MPC_MODEL_FDYN6, IRK != 0>(ls, b, it, last);
                }
                __syncthreads();
            }
        }
        if (s.hist_u0 && lane < 2) s.hist_u0[((size_t)step * B + b) * 2 + lane] = a.u0[(size_t)b * 2 + lane];
        if (s.hist_x0 && lane < 8) s.hist_x0[((size_t)step * B + b) * 8 + lane] = s.x0[(size_t)b * 8 + lane];
        if (s.hist_st && lane == 0) s.hist_st[(size_t)step * B + b] = a.status[b];
        if (s.hist_it && lane == 0) s.hist_it[(size_t)step * B + b] = a.qp_iter[b];
        if (s.freeze && s.x0[(size_t)b * 8] > s.lap_stop) {     // python/main.py:514-517: the lap is done
            if (s.active && lane == 0) s.active[b] = 0;
            __syncthreads();
        }
    }
}

}  // namespace

// This file is compiled THREE times (Makefile): QP_SET = 0 holds the all-hard instantiations (the reference's OCP), QP_SET = 1 the
// soft / track-row instantiations, QP_SET = 2 the persistent loop of the dynamic OCP models (all tables) -- same flags, same (default)
// scheduler.  Separate objects are separate device code images: the benchmarked kernels' image does not move when another set grows.  `make ilp` builds both again under
// LLVM's iterative ILP scheduler into the test artefact libihm2mpc_ilp.so (tests/test_gpu_configs.py compares the two builds).
#ifndef QP_SET
#error "compile with -DQP_SET=0 (all-hard instantiations), -DQP_SET=1 (soft / track-row instantiations) or -DQP_SET=2 (dynamic OCP models in the persistent loop)"
#endif
static size_t qp_lds_bytes(const ihm2mpc_handle *h)
{
    const size_t N = h->N, NS = h->NS;
    const int nck = h->path_on ? (h->alat_on ? 15 : 14) : 12;
    const int uni = h->uniform_H && h->uniform_CD;
    return sizeof(double) * (NS * (10 + 10 + 8 + 8 + 2 * nck + 10 + (h->path_on ? (h->alat_on ? 6 : 2) : 0)) + N * (8 + 4 + 16 + 8 + 8) + 136 + (uni ? 20 + (h->path_on ? 0 : 200 + 90) : 0));
}

static QpArgs qp_args(ihm2mpc_handle *h)
{
    QpArgs a;
    a.B = h->B; a.N = h->N; a.iter_max = h->cfg.ipm_iter_max; a.nslots = h->nslot_lane * 64; a.m_act = h->m_act;
    a.nslots_can = h->nslot_lane * 64;
    a.tol = h->cfg.ipm_tol; a.mu0 = h->cfg.ipm_mu0; a.tau0 = h->cfg.ipm_tau0;
    a.Hs = h->Hs; a.Gy = h->Gy; a.CD = h->CD; a.slot_lb = h->slot_lb; a.slot_ub = h->slot_ub; a.slot_kc = h->slot_kc;
    a.x = h->x; a.u = h->u; a.x0 = h->x0; a.yref = h->yref; a.yref_e = h->yref_e;
    a.pi = h->pi; a.lam = h->lam; a.res = h->res; a.qp_res = h->qp_res; a.u0 = h->u0; a.status = h->status; a.qp_iter = h->qp_iter;
    a.lin = h->lin; a.g = h->q_g; a.rg = h->q_rg; a.P = h->q_P; a.M = h->q_M + (size_t)QM_PAD * 64;
    a.slot_zw = h->slot_zw; a.slot_Zw = h->slot_Zw; a.slk = h->slk;
    a.track_id = h->track_id; a.widths = h->widths; a.car_L = h->car_L; a.car_W = h->car_W;
    a.lam_a = h->lam_a; a.slk_a = h->slk_a;
    a.symmetrize = (h->cfg.model == IHM2MPC_MODEL_FDYN6) ? 1 : 0;
    return a;
}

// n_steps control steps in one launch (k_steps).  Returns 0 launched, 1 the configuration has no persistent instantiation
// (the caller then runs ihm2mpc_step n_steps times, which gives the same results).  The all-hard tables are launched from the
// QP_SET = 0 object, the soft / track-row tables from the QP_SET = 1 object.
#if QP_SET == 0
int ihm2_launch_steps_soft(ihm2mpc_handle *h, int model, int M_sim, double s_target, int n_steps, int freeze, double lap_stop,
                           double *hist_u0, double *hist_x0, int32_t *hist_st, int32_t *hist_it);
int ihm2_launch_steps_dyn(ihm2mpc_handle *h, int model, int M_sim, double s_target, int n_steps, int freeze, double lap_stop,
                          double *hist_u0, double *hist_x0, int32_t *hist_st, int32_t *hist_it);
int ihm2_launch_steps(ihm2mpc_handle *h, int model, int M_sim, double s_target, int n_steps, int freeze, double lap_stop,
                      double *hist_u0, double *hist_x0, int32_t *hist_st, int32_t *hist_it)
#elif QP_SET == 1
int ihm2_launch_steps_soft(ihm2mpc_handle *h, int model, int M_sim, double s_target, int n_steps, int freeze, double lap_stop,
                           double *hist_u0, double *hist_x0, int32_t *hist_st, int32_t *hist_it)
#else
int ihm2_launch_steps_dyn(ihm2mpc_handle *h, int model, int M_sim, double s_target, int n_steps, int freeze, double lap_stop,
                          double *hist_u0, double *hist_x0, int32_t *hist_st, int32_t *hist_it)
#endif
{
    if (h->alat_on) return 1;       // the lateral-acceleration row has no instantiation of the persistent loop: launches per step (same results)
    const bool dyn = h->cfg.model != IHM2MPC_MODEL_FKIN6;
#if QP_SET == 0
    if (dyn) return ihm2_launch_steps_dyn(h, model, M_sim, s_target, n_steps, freeze, lap_stop, hist_u0, hist_x0, hist_st, hist_it);
#elif QP_SET == 1
    if (dyn) return 1;
#else
    if (!dyn || !(h->uniform_H && h->uniform_CD)) return 1;     // the dynamic models come with batch-shared tables only (the reference's OCP has them)
#endif
    const bool irk_plant = h->cfg.sim_integrator_type != IHM2MPC_INTEG_ERK;     // the plants by collocation (python/main.py:395-400: Radau IIA x M_sim)
    if (irk_plant && ihm2_upload_sim_irk_tab(h, M_sim)) return 1;
    const bool irk = h->cfg.integrator_type != IHM2MPC_INTEG_ERK;       // collocation step on the shooting intervals: batch-shared tables only
    if (irk && !(h->irk_tab && h->uniform_H && h->uniform_CD && (h->cfg.nlp_solver_type != IHM2MPC_SQP || !h->sqp_globalization || h->ls_phi))) return 1;
    const bool sqp = h->cfg.nlp_solver_type == IHM2MPC_SQP;
    if (sqp && !h->ls_x) return 1;        // the caller allocates the line-search buffers first
    const bool hard = !h->path_on && h->nsoft_lane == 0 && h->nslot_lane <= 10;
#if QP_SET == 0
    if (!hard) return ihm2_launch_steps_soft(h, model, M_sim, s_target, n_steps, freeze, lap_stop, hist_u0, hist_x0, hist_st, hist_it);
#elif QP_SET == 1
    if (hard) return 1;
#endif
    const size_t lds = qp_lds_bytes(h);
    if (lds > 160 * 1024) return 1;
    // the dynamic models' RK4 integrator parks its base sensitivities in the QP's LDS
    if (h->cfg.model != IHM2MPC_MODEL_FKIN6 && !irk && lds < (size_t)s_count(1) * 64 * sizeof(double)) return 1;
    QpArgs a = qp_args(h);
    StepArgs s;
    s.ocp_model = h->cfg.model;
    s.n_steps = n_steps; s.model = model; s.M_sim = M_sim; s.M = h->cfg.M; s.nknots = h->cfg.nknots; s.lap_wrap = h->lap_wrap ? 1 : 0;
    s.freeze = freeze; s.s_target = s_target; s.dt = h->cfg.dt; s.lap_stop = lap_stop;
    s.sqp_iters = sqp ? (h->cfg.nlp_solver_max_iter > 0 ? h->cfg.nlp_solver_max_iter : 1) : 0;
    s.s_ref = h->s_ref; s.kappa_ref = h->kappa_ref;
    s.x0 = h->x0; s.yref = h->yref; s.yref_e = h->yref_e; s.lin = h->lin;
    s.active = (freeze || h->active_set) ? h->active : nullptr;
    s.hist_u0 = hist_u0; s.hist_x0 = hist_x0; s.hist_st = hist_st; s.hist_it = hist_it;
    s.irk_tab = (const IrkTab *)h->irk_tab;
    s.sim_irk_tab = irk_plant ? (const IrkTab *)h->sim_irk_tab : nullptr;
    // every field of s is set: upload it (and the line search's block in the SQP mode)
    static_assert(sizeof(StepArgs) <= 32 * sizeof(double), "step_args holds 256 bytes");
    static_assert(sizeof(LsArgs) <= 64 * sizeof(double), "ls_args holds 512 bytes");
    // both blocks go through a pinned staging slot (two slots, used alternately) and are uploaded in stream order: the host does
    // not wait for the previous launch (run_steps(wait = false) enqueues in pieces while the host does other work)
    const int slot = (h->args_idx++) & 1;
    if (hipEventSynchronize(h->args_ev[slot]) != hipSuccess) return 1;          // the upload that last used this slot has been issued long ago
    char *stage = (char *)h->args_host[slot];
    std::memcpy(stage, &s, sizeof(StepArgs));
    if (hipMemcpyAsync(h->step_args, stage, sizeof(StepArgs), hipMemcpyHostToDevice, h->stream) != hipSuccess) return 1;
    if (sqp) {
        LsArgs ls_host = make_ls_args(h);
        if (irk) ls_host.phase = 3;        // the trial points' collocation rollouts are done in the loop, one step length at a time
        std::memcpy(stage + 512, &ls_host, sizeof(LsArgs));
        if (hipMemcpyAsync(h->ls_args, stage + 512, sizeof(LsArgs), hipMemcpyHostToDevice, h->stream) != hipSuccess) return 1;
    }
    if (hipEventRecord(h->args_ev[slot], h->stream) != hipSuccess) return 1;
    const StepArgs *sdev = (const StepArgs *)h->step_args;
    const LsArgs *ls = (const LsArgs *)h->ls_args;
    const int uni = h->uniform_H && h->uniform_CD;
    // one instantiation: slots per lane, soft slots per lane, track rows, batch-shared tables, SQP mode, collocation, dynamic model
#define LAUNCH_K(NS_, NO_, PT_, UN_, SQ_, IR_, DY_)                                                                                        \
    do {                                                                                                                                    \
        (void)hipFuncSetAttribute((const void *)k_steps<NS_, NO_, PT_, UN_, SQ_, IR_, DY_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_steps<NS_, NO_, PT_, UN_, SQ_, IR_, DY_>), dim3(h->B), dim3(64), lds, h->stream, sdev, a, ls);                \
    } while (0)
    // SQP mode and integrator at run time
#if QP_SET == 2
#define LAUNCH_STEPS(NS_, NO_, PT_, UN_)                                                                                                   \
    do {                                                                                                                                    \
        if (irk) { if (sqp) LAUNCH_K(NS_, NO_, PT_, 1, 1, 1, 1); else LAUNCH_K(NS_, NO_, PT_, 1, 0, 1, 1); }                                \
        else { if (sqp) LAUNCH_K(NS_, NO_, PT_, 1, 1, 0, 1); else LAUNCH_K(NS_, NO_, PT_, 1, 0, 0, 1); }                                    \
    } while (0)
#else
#define LAUNCH_STEPS(NS_, NO_, PT_, UN_)                                                                                                   \
    do {                                                                                                                                    \
        if (irk) { if (sqp) LAUNCH_K(NS_, NO_, PT_, 1, 1, 1, 0); else LAUNCH_K(NS_, NO_, PT_, 1, 0, 1, 0); }                                \
        else { if (sqp) LAUNCH_K(NS_, NO_, PT_, UN_, 1, 0, 0); else LAUNCH_K(NS_, NO_, PT_, UN_, 0, 0, 0); }                                \
    } while (0)
#endif
#if QP_SET != 1
    if (hard) {
        if (h->nslot_lane <= 5) { if (uni) LAUNCH_STEPS(5, 0, 0, 1); else LAUNCH_STEPS(5, 0, 0, 0); }
        else if (h->nslot_lane <= 8) { if (uni) LAUNCH_STEPS(8, 0, 0, 1); else LAUNCH_STEPS(8, 0, 0, 0); }
        else {      // long horizons (N <= 79 at 8 rows per stage): kinematic model, batch-shared tables
#if QP_SET == 0
            if (!uni) return 1;
            LAUNCH_STEPS(10, 0, 0, 1);
#else
            return 1;
#endif
        }
    }
#endif
#if QP_SET != 0
    // the soft / track-row tables: batch-shared Hessians and rows only (the reference's OCP has them)
    if (!hard) {
        if (!uni) return 1;
        const int per_lane = h->nslot_lane, nsoft = h->nsoft_lane;
        if (!h->path_on) {
            if (nsoft <= 2 && per_lane <= 8) LAUNCH_STEPS(8, 2, 0, 1);
            else if (nsoft <= 4 && per_lane <= 10) LAUNCH_STEPS(10, 4, 0, 1);
            else return 1;
        } else {
            // (the NSOFT of an instantiation = the leading ONE-SIDED entries of a lane, api.hip::rebuild_slots: an all-hard table takes NSOFT = 0)
            if (nsoft == 0) { if (per_lane <= 8) LAUNCH_STEPS(8, 0, 1, 1); else return 1; }
            else if (nsoft <= 3 && per_lane <= 8) LAUNCH_STEPS(8, 3, 1, 1);
            else if (nsoft <= 4 && per_lane <= 10) LAUNCH_STEPS(10, 4, 1, 1);
            else return 1;
        }
    }
#endif
#undef LAUNCH_STEPS
#undef LAUNCH_K
    return 0;
}

#if QP_SET != 2
#if QP_SET == 0
int ihm2_launch_qp_hard(ihm2mpc_handle *h)
#else
int ihm2_launch_qp_hard(ihm2mpc_handle *h);
int ihm2_launch_qp(ihm2mpc_handle *h)
#endif
{
#if QP_SET == 1
    if (!h->path_on && h->nsoft_lane == 0 && h->nslot_lane <= 10) return ihm2_launch_qp_hard(h);
#endif
    QpArgs a = qp_args(h);
    const int uni = h->uniform_H && h->uniform_CD;
    const size_t lds = qp_lds_bytes(h);
    if (lds > 160 * 1024) return 1;
    const int per_lane = h->nslot_lane, nsoft = h->nsoft_lane;
#define LAUNCH_QP(NS_, NO_, PT_)                                                                                          \
    do {                                                                                                                  \
        if (uni) {                                                                                                        \
            (void)hipFuncSetAttribute((const void *)k_qp_wave<NS_, NO_, PT_, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_qp_wave<NS_, NO_, PT_, 1>), dim3(h->B), dim3(64), lds, h->stream, a);                   \
        } else {                                                                                                          \
            (void)hipFuncSetAttribute((const void *)k_qp_wave<NS_, NO_, PT_, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_qp_wave<NS_, NO_, PT_, 0>), dim3(h->B), dim3(64), lds, h->stream, a);                   \
        }                                                                                                                 \
    } while (0)
#if QP_SET == 0
    // few instances (at most one per CU): four wavefronts per instance, slots from the 256-lane table
    if (h->block_qp && nsoft == 0 && !h->path_on && h->nslot_lane_blk >= 1 && h->nslot_lane_blk <= 2 && h->B <= h->n_cu && h->nslot_lane * 64 <= (h->N + 1) * 12) {
        a.slot_kc = h->slot_kc_blk; a.slot_lb = h->slot_lb_blk; a.slot_ub = h->slot_ub_blk; a.nslots = h->nslot_lane_blk * 256;
        if (uni) {
            (void)hipFuncSetAttribute((const void *)k_qp_block<2, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k_qp_block<2, 1, 4>), dim3(h->B), dim3(256), lds, h->stream, a);
        } else {
            (void)hipFuncSetAttribute((const void *)k_qp_block<2, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k_qp_block<2, 0, 4>), dim3(h->B), dim3(256), lds, h->stream, a);
        }
        return 0;
    }
    if (nsoft == 0 && per_lane <= 5) LAUNCH_QP(5, 0, 0);
    else if (nsoft == 0 && per_lane <= 8) LAUNCH_QP(8, 0, 0);
    else if (nsoft == 0 && per_lane <= 10) LAUNCH_QP(10, 0, 0);
    else return 2;
#else
    if (h->alat_on) {
        // track rows + the lateral-acceleration row (ready() has checked: kinematic model, track rows on, batch-shared tables)
        if (!uni || !h->path_on || nsoft > 4 || per_lane > 10 || (nsoft == 0 && per_lane > 8)) return 2;
        if (nsoft == 0) {       // all sides hard
            (void)hipFuncSetAttribute((const void *)k_qp_wave<8, 0, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k_qp_wave<8, 0, 2, 1>), dim3(h->B), dim3(64), lds, h->stream, a);
        } else {
            (void)hipFuncSetAttribute((const void *)k_qp_wave<10, 4, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k_qp_wave<10, 4, 2, 1>), dim3(h->B), dim3(64), lds, h->stream, a);
        }
    } else if (!h->path_on) {
        if (nsoft <= 2 && per_lane <= 8) LAUNCH_QP(8, 2, 0);
        else if (nsoft <= 4 && per_lane <= 10) LAUNCH_QP(10, 4, 0);
        else return 2;
    } else {
        if (nsoft == 0) { if (per_lane <= 8) LAUNCH_QP(8, 0, 1); else return 2; }
        else if (nsoft <= 3 && per_lane <= 8) LAUNCH_QP(8, 3, 1);
        else if (nsoft <= 4 && per_lane <= 10) LAUNCH_QP(10, 4, 1);
        else return 2;
    }
#endif
#undef LAUNCH_QP
    return 0;
}
#endif
