// ihm2mpc_internal.h -- handle layout and kernel launchers shared by the .hip translation units.
//
// Device data layout (DESIGN.md "Data layout in HBM"): instance-major, identical to the C-ABI's host
// layout, e.g. x[b][k][i] (B, N+1, 8).  The two hot kernels are shaped around it:
//   * linearize: one lane per (instance, interval), interval fastest -> a wavefront reads 64
//     consecutive 64-byte state rows (4 KB contiguous) and writes 64 consecutive 704-byte [A|B|b] records;
//   * QP: one wavefront per instance -> every per-stage record (704 B of [A|B|b], 512 B of P) is one
//     coalesced wave access, the iterate lives in LDS, multipliers/slacks in registers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ihm2mpc.h"

// Floating-point contraction.  The sources are compiled with -ffp-contract=on (Makefile) and switch to `fast` HERE, for everything that follows in
// the translation unit: a*b + c is then fused wherever the two operations meet, as before -- but by the `contract` flag on the operations, not by
// the back end's global switch (-ffp-contract=fast), which fuses every multiply-add pair it sees and cannot be turned off for a region.  The
// dynamic model's section of model.hpp turns contraction OFF: its results must not depend on what it is inlined into (NOTES.md R4.11).
#pragma clang fp contract(fast)

#define NX 8
#define NU 2
#define NZ 10
#define NY 12
#define NG 2
#define NH 2
#define NC 14   // two-sided constraint rows per stage: 8 state boxes, 2 input boxes, 2 general rows, 2 track rows
#define NLAM 28
#define MAX_SLOTS 640   // 10 per lane
#define QM_PAD 24    // >= 3 x ring depth of the vector / forward sweeps: their unclamped prefetch overshoots an instance by < 3 D rows
#define LIN_REC 96   // doubles per (instance, interval) linearisation record: A (64) | B (16) | b (8) | rb (8: the QP's dynamics residual, riccati_mfma.hpp)

struct ihm2mpc_comm;      // comm.hip: RCCL communicator + staging buffers of a one-process-per-GPU job

struct ihm2mpc_handle {
    ihm2mpc_config cfg;
    ihm2mpc_comm *comm;            // nullptr until ihm2mpc_comm_init
    int B, N, NS;
    int n_cu;                      // compute units of the device
    hipStream_t stream;
    hipStream_t stream2;           // plant + reference ramp of ihm2mpc_step run here, beside shift + linearisation
    hipEvent_t ev_fork, ev_join;
    hipEvent_t ev[4];
    bool tracks_set, weights_set, bounds_set;
    bool uniform_H, uniform_CD;    // stage Hessians / general rows identical for all k < N (QP kernel keeps them in LDS)

    // ---- shared problem data (device) ----
    double *s_ref, *kappa_ref;     // (ntracks, nknots)
    int32_t *track_id;             // (B)
    double *Hs;                    // (NS,10,10)  cost_scale * V'WV ; terminal: W_e padded with I
    double *Gy;                    // (NS,10,12)  cost_scale * V'W  ; terminal: W_e in the first 8x8
    double *lbx, *ubx;             // (NS,8)
    double *lbu, *ubu;             // (N,2)
    double *CD;                    // (N,2,10)  general rows [C D]
    double *lg, *ug;               // (N,2)
    // compact table of the constraint slots that have at least one finite side
    // -- laid out for the QP kernel: entry lane + 64 r belongs to lane `lane`; a lane's soft slots come first --
    int nslots, m_act;             // m_act: number of one-sided inequality pairs (finite sides + one per soft slack)
    int nslot_lane, nsoft_lane;    // slots per lane / leading one-sided entries per lane (the NSOFT of the instantiation that takes the table; 0: all-hard)
    bool slots_fit;                // false: the rows set so far fit no instantiation (reported by the next solve: a later setter may still change them)
    int32_t *slot_kc;              // (nslot_lane*64) stage * 16 + row, -1 = padding
    int32_t *slot_kc_blk;          // the same rows spread over 256 lanes (k_qp_block: four wavefronts per instance), all-hard tables only
    double *slot_lb_blk, *slot_ub_blk;
    int nslot_lane_blk;            // 0: no such table (soft sides present)
    bool block_qp;                 // use k_qp_block for batches of at most one instance per CU (IHM2MPC_BLOCK_QP=0 turns it off)
    double *slot_lb, *slot_ub;     // raw bounds, +-inf if that side is absent (soft slots are one-sided)
    double *slot_zw, *slot_Zw;     // slack cost zw s + 1/2 Zw s^2 of a soft slot; Zw < 0 = hard slot
    // host copies the table is rebuilt from (set_bounds / set_soft may come in either order)
    double *host_lb, *host_ub;     // (NS*NC) per (stage, row), +-inf = absent
    double *host_sz, *host_sZ;     // (NS*NLAM) per one-sided constraint: NC lower then NC upper
    // nonlinear track-boundary rows (rows 12, 13 of the stages 1..N)
    int path_on;
    double car_L, car_W, lh[NH], uh[NH];
    double *widths;                // (ntracks, 2) = (w_R, w_L), device
    // lateral-acceleration row of the kinematic constraint set (row 14 of the stages 1..N-1; ihm2mpc_set_alat_constraint)
    int alat_on;
    double alat_lb, alat_ub, alat_sz[2], alat_sZ[2];      // bounds (+-inf = absent), slack penalties of the lower / upper side (sZ < 0 = hard)
    // Cartesian side (ROS stack): centre-line geometry per track, Cartesian plant state and projection guess per instance
    bool geometry_set;
    double *X_ref, *Y_ref, *phi_ref;   // (ntracks, nknots)
    double *xc;                    // (B,8) (X, Y, phi, v_x, v_y, r, T, delta)
    double *s_guess;               // (B)

    // ---- per-instance state, instance-major ----
    double *x;      // (B,NS,8)
    double *u;      // (B,N,2)
    double *x0;     // (B,8)
    double *yref;   // (B,N,12)
    double *yref_e; // (B,8)
    double *pi;     // (B,NS,8)
    double *lam;    // (B,NS,28)
    double *slk;    // (B,NS,28) slack values of the soft sides after the last QP (0 for hard sides)
    double *lam_a, *slk_a;   // (B,NS,2) multipliers and slack values of the lateral-acceleration row: lower, upper side
    double *res;    // (B,4)
    double *ls_phi; // (n_alpha, B, N, 8) IRK rollouts at the trial points of the line search (SQP mode with the IRK integrator)
    int ls_nalpha;
    void *irk_tab;       // device copy of the OCP integrator's collocation tableau (irk_body.hpp: IrkTab), nullptr for ERK
    void *sim_irk_tab;   // the same for the plant steps of the persistent loop (step dt / sim_irk_M), allocated on first use
    int sim_irk_M;
    int32_t *ls_pending; // (B) instances whose line search goes past the first rollouts (two-launch ladder)
    double *dyn10;  // (B,35) staging of the fdyn10 plant: x (15), u (5), x_next (15); allocated on first use
    double *qp_res; // (B,4) KKT residuals of the QP at its returned point, relative to the scales of its tolerances
    int32_t *status, *qp_iter;   // (B)
    int32_t *active;             // (B) plant mask of the device-resident closed loop (nullptr-equivalent while !active_set)
    bool active_set;
    bool freeze_armed;           // a run_steps(freeze) call initialised the mask: later calls keep what the device made of it
    bool lap_wrap;               // prepare_step / step move cars that passed s = L back by one lap first
    double *u0;     // (B,2) first control of the last solve

    double *lin;    // (B,N,96) linearisation records [A | B | b | rb], then B spare records (the kinematic plant's by-product)
    // ---- QP workspace in HBM/L2 (everything else of the QP lives in LDS / registers) ----
    double *q_g;    // (B,NS,10) QP gradient
    double *q_rg;   // (B,NS,10) stationarity residual of the interior-point iterate (follows the step between two evaluations from the data)
    double *q_P;    // (B,NS,64) Riccati matrices of the current factorisation
    double *q_M;    // (QM_PAD + B*N + QM_PAD, 64) closed-loop matrices A - B K (row-major; the vector recursion reads them
                    // transposed), padded at both ends: the sweeps' prefetch rings run QM_PAD rows past an instance unclamped
    double *scratch;   // (B, 3*8) plant scratch

    // ---- SQP mode (cfg.nlp_solver_type == IHM2MPC_SQP): convergence test + merit line search, kernels_sqp.hip ----
    int sqp_globalization, sqp_use_suff, sqp_full_step_dual;   // globalization: 0 FIXED_STEP, 1 MERIT_BACKTRACKING
    double sqp_alpha_min, sqp_alpha_red, sqp_eps, sqp_tol[4];
    double *Wd;                     // (N,12,12) then W_e (8,8): the merit function evaluates the cost from the weights themselves
    double *st_lb, *st_ub;          // (NS,NC) device copies of host_lb / host_ub
    double *st_sz, *st_sZ;          // (NS,NLAM) device copies of host_sz / host_sZ
    // allocated by the first SQP solve: the iterate the QP was built at, merit weights, per-solve bookkeeping
    double *ls_x, *ls_u, *ls_pi, *ls_lam, *ls_slk, *ls_wpi, *ls_wlam, *ls_alpha;
    int32_t *ls_done, *ls_status, *ls_iter, *ls_qp_acc;
    double *step_args;              // device copy of the persistent loop's own argument block (256 B), allocated with the handle
    double *ls_args;                // device copy of the line search's argument block for the persistent loop (512 B)
    void *args_host[2];             // pinned staging of both blocks (1 KB each), used alternately
    hipEvent_t args_ev[2];          // recorded after a slot's upload: the slot is free again once it has passed
    int args_idx;

    // ---- history of ihm2mpc_run_steps, grown on demand ----
    size_t hist_cap;                // steps the buffers hold
    double *hist_u0, *hist_x0;      // (steps,B,2), (steps,B,8)
    int32_t *hist_st, *hist_it;     // (steps,B)
};

// --- launchers (each defined in one .hip file) ---
// mode: bit 0 = reference ramp (needs x0), bit 1 = warm-start shift
void ihm2_launch_prepare(ihm2mpc_handle *h, double s_target, int mode, hipStream_t stream);
void ihm2_launch_build_tracks(ihm2mpc_handle *h, int max_seg, const int32_t *nseg, const double *cX, const double *cY, double *work);
void ihm2_launch_track_fit(ihm2mpc_handle *h, int max_pts, const int32_t *npts, const double *xy, double curv_weight, double *work, double *cX, double *cY,
                           int32_t *fail_flag);
void ihm2_launch_sim_dyn10(ihm2mpc_handle *h, int M, const double *x, const double *u, double *xn, hipStream_t stream);
void ihm2_launch_sim_dyn10_irk(ihm2mpc_handle *h, int integ, int M, int newton_iter, const double *x, const double *u, double *xn, hipStream_t stream);
void ihm2_launch_wrap_lap(ihm2mpc_handle *h);
void ihm2_launch_init_guess(ihm2mpc_handle *h, double v_ref_scale, int only_failed);
void ihm2_launch_linearize(ihm2mpc_handle *h);
// kernels_irk.hip: the collocation integrators (cfg.integrator_type / cfg.sim_integrator_type != IHM2MPC_INTEG_ERK)
void ihm2_launch_linearize_irk(ihm2mpc_handle *h);
int ihm2_upload_sim_irk_tab(ihm2mpc_handle *h, int M_sim);      // 0 ok (h->sim_irk_tab holds the plant tableau for M_sim steps)
int ihm2_upload_irk_tab(ihm2mpc_handle *h);      // (re)builds h->irk_tab for the configured integrator and step; 0 ok
void ihm2_launch_rollout_irk(ihm2mpc_handle *h, int j_begin, int j_end, double *phi, const int32_t *pending);
void ihm2_launch_sim_irk(ihm2mpc_handle *h, int model, int M_sim, const double *x, const double *u, double *xn, hipStream_t stream,
                         const int32_t *active);
void ihm2_launch_copy_iterate(ihm2mpc_handle *h);   // x, u, pi, lam, slk -> ls_x, ls_u, ls_pi, ls_lam, ls_slk
void ihm2_launch_line_search(ihm2mpc_handle *h, int it, int last, int phase = 0, int j_limit = 0);
int ihm2_launch_qp(ihm2mpc_handle *h);
// the persistent per-instance loop (kernels_qp.hip); returns 1 if the configuration has no instantiation of it
int ihm2_launch_steps(ihm2mpc_handle *h, int model, int M_sim, double s_target, int n_steps, int freeze, double lap_stop,
                      double *hist_u0, double *hist_x0, int32_t *hist_st, int32_t *hist_it);   // returns non-zero if the problem does not fit the kernel's limits
void ihm2_launch_sim(ihm2mpc_handle *h, int model, int M_sim, const double *x, const double *u, double *xn, hipStream_t stream,
                     const int32_t *active);
void ihm2_launch_sim_cart(ihm2mpc_handle *h, int model, int M, double dt, int n_steps, double v_dyn, const double *x, const double *u,
                          double *xn, hipStream_t stream);
void ihm2_launch_project(ihm2mpc_handle *h, double s_tol, const double *xc, double *s_guess, double *xf, hipStream_t stream);
