/*
 * ihm2mpc.h -- C ABI of libihm2mpc.so: batched SQP-RTI solver for the ihm2 path-parametric
 * bicycle NMPC on AMD MI355X (gfx950).  Plain C: opaque handle, pointers and sizes only.
 *
 * This is the drop-in boundary for the reference's NMPC hot path.  Each entry point replaces an
 * acados call the reference makes (file:line relative to tudoroancea/ihm2):
 *
 *   ihm2mpc_create            <- AcadosOcpSolver(ocp, json_file=...)            python/mpc.py:111-113
 *                                ihm2_fkin6_acados_create_capsule/_create       src/ihm2/src/mpc_control_node.cpp:274-284
 *   ihm2mpc_free              <- ihm2_fkin6_acados_free/_free_capsule           mpc_control_node.cpp:398-412
 *   ihm2mpc_set_tracks        <- solver.set(i, "p", p) for all stages           python/main.py:249-252
 *                                ihm2_fkin6_acados_update_params                mpc_control_node.cpp:360-364
 *   ihm2mpc_set_weights       <- solver.cost_set(i, "W", W)                     python/main.py:253-295; mpc_control_node.cpp:287-293
 *   ihm2mpc_set_bounds        <- ocp.constraints.* / constraints_set(i,"lbx"..) python/mpc.py:79-99; dpc/main.py:226-227,259-261
 *   ihm2mpc_set_x0            <- solver.set(0,"lbx",x); set(0,"ubx",x)          python/main.py:299-300; mpc_control_node.cpp:177-179
 *   ihm2mpc_set_yref(_e)      <- solver.set(j,"yref",..)                        python/main.py:303-314; mpc_control_node.cpp:183-186
 *   ihm2mpc_set_x / _set_u    <- solver.set(j,"x"/"u",..) (warm start)          python/main.py:317-322
 *   ihm2mpc_prepare_step      <- the whole of python/main.py:303-322 on device (reference ramp + shift)
 *   ihm2mpc_solve             <- solver.solve()                                 python/main.py:325; mpc_control_node.cpp:189
 *   ihm2mpc_set_sqp_options   <- ocp.solver_options.globalization / alpha_min / ... / nlp_solver_tol_*   python/main.py:230-237
 *   ihm2mpc_get_sqp_stats     <- solver.get_stats("sqp_iter") / ("alpha") (acados)
 *   ihm2mpc_get_x/_u/_u0      <- solver.get(i,"x"/"u")                          python/main.py:331-334; mpc_control_node.cpp:202-206
 *   ihm2mpc_get_status        <- return value of solve()                        python/main.py:325-328; dpc/main.py:287-293
 *   ihm2mpc_get_residuals     <- solver.get_stats("residuals") (acados)
 *   ihm2mpc_sim_step          <- AcadosSimSolver.simulate(x,u)                  python/main.py:476-502; python/sim.py:9-25
 *   ihm2mpc_compute_control   <- IHM2Controller.compute_control(x) as one call            python/main.py:297-334
 *   ihm2mpc_step              <- one iteration of the MiL loop (plant + compute_control)  python/main.py:476-517
 *   ihm2mpc_run_steps         <- n iterations of that loop in one launch                  python/main.py:448-517
 *   ihm2mpc_set_soft          <- ocp.constraints.idxsbx/idxsg/idxsh, cost.zl..Zu         old/generate_acaods_interface.py:380-449
 *   ihm2mpc_set_path_constraints <- model.con_h_expr (track rows), constraints.lh/uh     old/generate_acaods_interface.py:191-212,411-449
 *   ihm2mpc_set_track_geometry, ihm2mpc_project <- Track(csv), Track::project + Frenet states
 *                                                   src/ihm2/src/common/tracks.cpp:132-288; mpc_control_node.cpp:142-157
 *   ihm2mpc_sim_step_cart     <- ihm2_kin6/dyn6_acados_sim_solve + switch + clamp        src/ihm2/src/sim_node.cpp:197-257
 *
 * Conventions
 *   x = (s, n, psi, v_x, v_y, r, T, delta), u = (u_T, u_delta)   (python/models.py:236-245)
 *   All host arrays are C-contiguous, instance-major: x (B,N+1,8), u (B,N,2), yref (B,N,12) ...
 *   Every setter copies in, every getter copies out; no caller pointer is kept after return.
 *   A handle owns its device memory and its HIP streams (one, plus a helper stream inside ihm2mpc_step); it is not thread-safe; distinct handles
 *   are independent (one per device for multi-GPU sharding).
 *   Return value: 0 = ok, < 0 = API misuse or HIP error (text in ihm2mpc_last_error()).
 *   Per-instance solver status (acados codes): 0 success, 1 NaN/failure, 2 max iterations (SQP
 *   mode), 3 min step, 4 QP failure.  The reference accepts {0, 2} (python/main.py:326).
 */
#ifndef IHM2MPC_H
#define IHM2MPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IHM2MPC_NX 8
#define IHM2MPC_NU 2
#define IHM2MPC_NY 12
#define IHM2MPC_NYE 8
#define IHM2MPC_NG 2
#define IHM2MPC_NLAM 28 /* multipliers per stage: lower (8 bx, 2 bu, 2 g, 2 h) then upper (8, 2, 2, 2) */
#define IHM2MPC_NMAX 128 /* structural limit; the QP kernel also needs <= 640 constraint slots (two-sided rows with a finite
                            side, soft sides count separately): N <= 79 with the reference's 8 rows per stage */

#define IHM2MPC_MODEL_FKIN6 0 /* python/models.py:232-307 */
#define IHM2MPC_MODEL_FDYN6 1 /* python/models.py:455-606 */
/* fdyn6 with every wheel's lateral force on its OWN slip angle.  The reference crosses them (F_lat_FL uses alpha_RR ...,
 * python/models.py:543-546, quirk Q3); as written the model is open-loop unstable (yaw eigenvalue +34 1/s at 10 m/s) */
#define IHM2MPC_MODEL_FDYN6U 2

/* interior-point step: fraction of the distance to the boundary a step may take (primal and dual step lengths separately).  Round 4 tried
 * 0.9999: -3.6 % iterations in the oracle's closed loop, but instances of the benchmark batch start to fail and GPU / oracle / the two compiler
 * schedulers drift apart on marginal QPs (six GPU tests); 0.995 stays (NOTES.md R4) */
#define IHM2MPC_IPM_STEP_FRACTION 0.995

#define IHM2MPC_SQP_RTI 0 /* old/generate.py:21 */
#define IHM2MPC_SQP 1     /* python/main.py:230 */
#define IHM2MPC_FIXED_STEP 0         /* full steps */
#define IHM2MPC_MERIT_BACKTRACKING 1 /* python/main.py:237 */

/* integrators (AcadosOcpOptions.integrator_type / collocation_type, AcadosSimOpts): ERK = classical RK4 x M sub-steps; IRK = 4-stage
 * collocation x M steps, 3 Newton iterations per step from K = 0 with a fresh Jacobian each (acados' defaults), forward
 * sensitivities by the implicit-function theorem; GL4 = GAUSS_LEGENDRE (acados' default, python/main.py:234-236), RADAU4 =
 * GAUSS_RADAU_IIA (python/main.py:395-400, python/sim.py:28-33).  Both NLP solver types run on either integrator (the live options of
 * python/main.py:227-238 are SQP + MERIT_BACKTRACKING + IRK); the persistent loop (ihm2mpc_run_steps in one launch) takes RK4 or IRK on the
 * shooting intervals (batch-shared tables for IRK) of the kinematic and the dynamic OCP models, with RK4 or Radau IIA plants, and falls back to
 * launches per step otherwise. */
#define IHM2MPC_INTEG_ERK 0
#define IHM2MPC_INTEG_IRK_GL4 1
#define IHM2MPC_INTEG_IRK_RADAU4 2
#define IHM2MPC_IRK_NEWTON_ITER 3

typedef struct ihm2mpc_handle ihm2mpc_handle;

typedef struct ihm2mpc_config {
    int32_t batch;          /* B: independent MPC instances on this device */
    int32_t N;              /* shooting intervals (python/main.py:183: Nf = 40) */
    int32_t M;              /* integrator steps per interval (sim_method_num_steps): RK4 sub-steps, >= 18 at dt = 0.05 for stability (ERK);
                             * 1 for IRK (python/main.py:236) */
    int32_t model;          /* IHM2MPC_MODEL_* of the OCP */
    int32_t ntracks;        /* number of track tables */
    int32_t nknots;         /* knots per table (python/motion_planning.py:25,402-428: 3*500) */
    int32_t device;         /* HIP device ordinal */
    int32_t nlp_solver_type;     /* IHM2MPC_SQP_RTI or IHM2MPC_SQP */
    int32_t nlp_solver_max_iter; /* SQP mode: iterations per solve() (python/main.py:231) */
    int32_t ipm_iter_max;   /* interior-point iteration cap */
    double dt;              /* interval length (python/main.py:184) */
    double cost_scale_stage;/* factor on the stage cost terms (acados: the time step) */
    double ipm_tol;         /* relative tolerance of the QP solver */
    double ipm_mu0;         /* initial barrier parameter factor */
    double ipm_tau0;        /* initial slack floor */
    double nlp_tol;         /* SQP mode: KKT tolerance */
    int32_t integrator_type;     /* IHM2MPC_INTEG_* of the shooting intervals (python/main.py:234: "IRK"; old/generate.py:23: "ERK") */
    int32_t sim_integrator_type; /* IHM2MPC_INTEG_* of the plant steps (python/main.py:395-400: IRK, GAUSS_RADAU_IIA, 4 stages) */
} ihm2mpc_config;

const char *ihm2mpc_last_error(void);
const char *ihm2mpc_version(void);

int ihm2mpc_create(const ihm2mpc_config *cfg, ihm2mpc_handle **out);
int ihm2mpc_free(ihm2mpc_handle *h);
int ihm2mpc_synchronize(ihm2mpc_handle *h);
/* the handle's hipStream_t, as void* (for callers that enqueue their own work behind a solve) */
int ihm2mpc_get_stream(ihm2mpc_handle *h, void **stream);

/* ---- problem data shared by the whole batch ---- */
int ihm2mpc_set_tracks(ihm2mpc_handle *h, const double *s_ref, const double *kappa_ref); /* (ntracks,nknots) each */
/* The same tables (and the centre-line geometry of ihm2mpc_set_track_geometry) built ON THE DEVICE from the cubic spline coefficients of
 * the centre lines -- python/motion_planning.py:139-289 (segment lengths on 100 points, uniform arc-length resampling, heading, curvature),
 * :345-399 (offline_motion_plan: heading offset -asin(l_R kappa), lap length) and :402-428 (three laps side by side): nknots = 3 x samples
 * per lap.  coeffs_X, coeffs_Y: (ntracks, max_seg, 4), segment j of track t in [c0, c1, c2, c3] of X(t) = c0 + c1 t + c2 t^2 + c3 t^3,
 * t in [0, 1]; nseg (ntracks): segments of each track (the rest of its rows is ignored).  The closed-spline fit that produces the
 * coefficients (python/motion_planning.py:28-124: one small equality-constrained least-squares problem per track): ihm2mpc_fit_tracks, or any host fit. */
int ihm2mpc_build_tracks(ihm2mpc_handle *h, int32_t max_seg, const int32_t *nseg, const double *coeffs_X, const double *coeffs_Y);
/* The closed cubic-spline fit itself ON THE DEVICE -- replaces fit_spline of python/motion_planning.py:28-124 (there: qpsolvers / proxqp on
 * min 1/2 p'Pp + q'p s.t. Ap = 0; here: its KKT system, one workgroup per track, sparse Gaussian elimination with partial pivoting).
 * xy (ntracks, max_pts, 2): centre-line points of every track, the first npts[t] rows used (closed path: the last point is NOT the first
 * again); curv_weight: weight of the curvature term (offline_motion_plan uses 2.0, :358); out coeffs_X, coeffs_Y (ntracks, max_pts, 4), host:
 * feed them to ihm2mpc_build_tracks.  3 <= npts[t] <= max_pts <= 182. */
int ihm2mpc_fit_tracks(ihm2mpc_handle *h, int32_t max_pts, const int32_t *npts, const double *xy, double curv_weight, double *coeffs_X,
                       double *coeffs_Y);
/* read the tables back, (ntracks, nknots) each; any pointer may be NULL */
int ihm2mpc_get_tracks(ihm2mpc_handle *h, double *s_ref, double *kappa_ref, double *X_ref, double *Y_ref, double *phi_ref);
int ihm2mpc_set_track_id(ihm2mpc_handle *h, const int32_t *track_id);                     /* (B) */
int ihm2mpc_set_weights(ihm2mpc_handle *h, const double *W, const double *W_e);           /* (N,12,12), (8,8) */
/* lbx/ubx (N+1,8) [row 0 unused], lbu/ubu (N,2), C (N,2,8), D (N,2,2), lg/ug (N,2); +-inf = absent */
int ihm2mpc_set_bounds(ihm2mpc_handle *h, const double *lbx, const double *ubx, const double *lbu,
                       const double *ubu, const double *C, const double *D, const double *lg,
                       const double *ug);
/* SOFT constraint sides (AcadosOcpConstraints idxsbx / idxsbx_e / idxsg with AcadosOcpCost zl, zu, Zl, Zu --
 * declared by the reference's OCP class, python/mpc.py:58-90 uses hard sides only): soft_z, soft_Z (N+1,28) per
 * one-sided constraint, 14 lower sides [x(8) u(2) g(2) h(2)] then 14 upper sides.  A side with soft_Z >= 0 carries a
 * slack s >= 0 with cost soft_z*s + 1/2*soft_Z*s^2; soft_Z < 0 = hard.  NULL, NULL = all hard (the default). */
int ihm2mpc_set_soft(ihm2mpc_handle *h, const double *soft_z, const double *soft_Z);
/* Nonlinear track-boundary rows (model.con_h_expr / con_h_expr_e of old/generate_acaods_interface.py:191-212), rows 12
 * and 13 of the stages 1..N:
 *     h_R = n - 1/2 L sin|psi| + 1/2 W cos(psi) - w_R ,   h_L = -n + 1/2 L sin|psi| + 1/2 W cos(psi) - w_L ,
 * lh <= h <= uh (2 entries each; |bound| >= 1e20 = absent; the reference writes lh = -1e3, uh = 0, :411-449), linearised at
 * the iterate by every RTI step.  L, W = car length / width (python/constants.py:57-58); widths (ntracks,2) = (w_R, w_L),
 * constant along a track as the reference's motion plan tiles them (python/motion_planning.py:385-386).
 * enable = 0 removes the rows (the other arguments are then ignored and may be NULL). */
int ihm2mpc_set_path_constraints(ihm2mpc_handle *h, int32_t enable, double car_length, double car_width,
                                 const double *widths, const double *lh, const double *uh);
/* The lateral-acceleration row of the KINEMATIC model's nonlinear constraint set -- old/generate_acaods_interface.py:198-209
 * (`[right, left, T_dot, delta_dot] + ([] if is_dynamic else [a_lat])`), bounds :52-53, :424, :433 (-5 / +5 m/s^2), definition :266-271 and
 * old/scripts/gen_mpc.py:182-184, with the forces and the slip angle of python/models.py:255-263:
 *     a_lat = (-F_Rx sin(beta) + F_Fx sin(delta - beta)) / m + (v_x^2 + v_y^2) sin(beta) / l_R ,   a_lat_min <= a_lat(x_k) <= a_lat_max
 * on the stages 1..N-1 (x_0 is fixed; it is no terminal row: con_h_expr_e, :209-212), linearised at the iterate by every RTI step: a
 * fifteenth row of a stage, with non-zeros in (v_x, v_y, T, delta).  soft_z, soft_Z (2 each: lower side, upper side; NULL = hard): slack
 * penalties as ihm2mpc_set_soft (the reference softens every h row, :380-395).  Needs the kinematic OCP model, the track rows
 * (ihm2mpc_set_path_constraints enabled), SQP_RTI and stage-independent weights; ihm2mpc_run_steps then launches per step.
 * The row's multipliers and slack values do not change the 28-column layout of the other rows: (B,N+1,2) = (lower, upper) arrays of
 * their own, zeroed by ihm2mpc_set_multipliers(.., NULL) / ihm2mpc_set_slacks(NULL) like the others.
 * enable = 0 removes the row. */
int ihm2mpc_set_alat_constraint(ihm2mpc_handle *h, int32_t enable, double a_lat_min, double a_lat_max, const double *soft_z,
                                const double *soft_Z);
int ihm2mpc_set_alat_multipliers(ihm2mpc_handle *h, const double *lam, const double *slk); /* (B,N+1,2) each; NULL = zero */
int ihm2mpc_get_alat_multipliers(ihm2mpc_handle *h, double *lam, double *slk);             /* (B,N+1,2) each; either may be NULL */

/* ---- per-instance data ---- */
int ihm2mpc_set_x0(ihm2mpc_handle *h, const double *x0);         /* (B,8) */
int ihm2mpc_set_x(ihm2mpc_handle *h, const double *x);           /* (B,N+1,8) */
int ihm2mpc_set_u(ihm2mpc_handle *h, const double *u);           /* (B,N,2) */
int ihm2mpc_set_yref(ihm2mpc_handle *h, const double *yref);     /* (B,N,12) */
int ihm2mpc_set_yref_e(ihm2mpc_handle *h, const double *yref_e); /* (B,8) */
int ihm2mpc_set_multipliers(ihm2mpc_handle *h, const double *pi, const double *lam); /* (B,N+1,8), (B,N+1,28); NULL = zero */

/* one stage of one instance (the AcadosOcpSolver.set/get call shape); field is one of
 * "x","u","yref","yref_e","lbx","ubx" (stage 0 only: both set x0),"pi","lam" */
int ihm2mpc_set_stage(ihm2mpc_handle *h, int32_t instance, int32_t stage, const char *field,
                      const double *value, int32_t n);
int ihm2mpc_get_stage(ihm2mpc_handle *h, int32_t instance, int32_t stage, const char *field,
                      double *value, int32_t n);

/* ---- the hot path ---- */
/* initial guess: roll the model out from x0 under a Stanley-type feedback (python/main.py:99-163) */
int ihm2mpc_init_guess(ihm2mpc_handle *h, double v_ref_scale);
/* the same rollout for the instances whose last solve failed (status other than 0 and 2) only, multipliers cleared: a failed instance keeps
 * its iterate, which can keep it infeasible for ever; the reference stops its single loop instead
 * (python/main.py:326-328).  Call between solve() and the next prepare_step(). */
int ihm2mpc_reinit_failed(ihm2mpc_handle *h, double v_ref_scale);
/* reference ramp yref_j = [s0 + s_target*j/N, 0..], yref_e = [s0 + s_target, 0..] from the current
 * x0, and warm-start shift of (x,u) -- python/main.py:303-322 -- entirely on device */
int ihm2mpc_prepare_step(ihm2mpc_handle *h, double s_target);
/* n_iter RTI iterations (n_iter <= 0: the configured nlp_solver_max_iter / 1 for SQP_RTI).
 * With nlp_solver_type IHM2MPC_SQP (python/main.py:230-237) every iteration first tests the four KKT residuals of the iterate
 * against the tolerances -- a converged instance gets status 0 and is left alone -- and the QP step is scaled by the line
 * search chosen with ihm2mpc_set_sqp_options; an instance still iterating after n_iter QPs gets status 2 (ACADOS_MAXITER),
 * a QP failure ends its solve with status 1 / 4.  get_qp_iter then reports the interior-point iterations of all its QPs. */
int ihm2mpc_solve(ihm2mpc_handle *h, int32_t n_iter);
/* SQP mode only.  globalization IHM2MPC_FIXED_STEP (default) or IHM2MPC_MERIT_BACKTRACKING: backtracking on the l1 merit
 * function  cost + sum w |dynamics defect| + sum w max(0, constraint violation)  with weights following the QP multipliers
 * (|mult|, then max(|mult|, (w + |mult|)/2)); trial steps alpha = 1, alpha_reduction, alpha_reduction^2, ... >= alpha_min,
 * accepted on plain decrease or, with use_sufficient_descent, on m(alpha) - m(0) <= eps_sufficient_descent alpha D;
 * full_step_dual keeps the QP multipliers instead of moving them by alpha.  tol (4): stat, eq, ineq, comp, NULL keeps the
 * current ones (cfg.nlp_tol for all four).  Defaults: 0.05, 0.7, 1e-4, 0, 0 -- acados' defaults. */
int ihm2mpc_set_sqp_options(ihm2mpc_handle *h, int32_t globalization, double alpha_min, double alpha_reduction,
                            double eps_sufficient_descent, int32_t use_sufficient_descent, int32_t full_step_dual,
                            const double *tol);
/* QP solves made (B) and the last step length (B) of the last SQP-mode solve; either pointer may be NULL */
int ihm2mpc_get_sqp_stats(ihm2mpc_handle *h, int32_t *sqp_iter, double *alpha);
/* phases of solve(), exposed for parity tests and profiling */
int ihm2mpc_linearize(ihm2mpc_handle *h);
int ihm2mpc_get_linearization(ihm2mpc_handle *h, double *A, double *Bm, double *b); /* (B,N,8,8),(B,N,8,2),(B,N,8) */

int ihm2mpc_get_x(ihm2mpc_handle *h, double *x);
int ihm2mpc_get_u(ihm2mpc_handle *h, double *u);
int ihm2mpc_get_u0(ihm2mpc_handle *h, double *u0);              /* (B,2) */
int ihm2mpc_get_status(ihm2mpc_handle *h, int32_t *status);     /* (B) */
int ihm2mpc_get_qp_iter(ihm2mpc_handle *h, int32_t *qp_iter);   /* (B) */
int ihm2mpc_get_residuals(ihm2mpc_handle *h, double *res);      /* (B,4): stat, eq, ineq, comp */
/* (B,4): inf-norm KKT residuals of the QP (stationarity, dynamics, inequalities, complementarity) at the point the interior-point
 * iteration returned, each divided by the scale its tolerance is relative to: <= ipm_tol for status 0.  acados: the qp_res of
 * solver.get_stats("qp_res_*") / print_statistics() */
int ihm2mpc_get_qp_residuals(ihm2mpc_handle *h, double *res);
int ihm2mpc_get_multipliers(ihm2mpc_handle *h, double *pi, double *lam);
/* slack values the next SQP-mode solve starts from (its line search walks from them to the QP's); NULL = zeros */
int ihm2mpc_set_slacks(ihm2mpc_handle *h, const double *sl);     /* (B,N+1,28) */
int ihm2mpc_get_slacks(ihm2mpc_handle *h, double *sl);         /* (B,N+1,28) slack of each soft side after the last QP */
/* milliseconds of the last solve(): [0] total, [1] linearize, [2] qp+update (HIP events) */
int ihm2mpc_get_timings(ihm2mpc_handle *h, double *ms, int32_t n);

/* ---- device-pointer variants (zero-copy closed loop, RCCL gather of results) ----
 * dptr is device memory on the handle's device, SAME (instance-major) layout as the host variant */
int ihm2mpc_set_x0_device(ihm2mpc_handle *h, const void *dptr);
int ihm2mpc_get_u0_device(ihm2mpc_handle *h, void *dptr);
int ihm2mpc_get_x_device(ihm2mpc_handle *h, void *dptr);
int ihm2mpc_get_u_device(ihm2mpc_handle *h, void *dptr);
int ihm2mpc_get_status_device(ihm2mpc_handle *h, void *dptr);

/* ---- plant step (closed-loop MiL): x_next = RK4 x M_sim over dt of `model` under u ---- */
int ihm2mpc_sim_step(ihm2mpc_handle *h, int32_t model, int32_t M_sim, const double *x,
                     const double *u, double *x_next); /* host (B,8),(B,2) -> (B,8) */
/* on device: x0 <- plant(x0, u0 of the last solve); model = -1: kin/dyn switch of python/main.py:482-489 (-2: the same
 * switch with IHM2MPC_MODEL_FDYN6U as the dynamic model) */
int ihm2mpc_sim_advance(ihm2mpc_handle *h, int32_t model, int32_t M_sim);
int ihm2mpc_get_x0(ihm2mpc_handle *h, double *x0);
/* one iteration of the MiL loop (python/main.py:476-517) on the device: sim_advance(model, M_sim), prepare_step(s_target) and
 * one RTI iteration.  Same results as the three calls; the plant step and the reference ramp run beside the warm-start
 * shift and the linearisation (they only meet in the QP), which hides the plant's latency. */
int ihm2mpc_step(ihm2mpc_handle *h, int32_t model, int32_t M_sim, double s_target);
/* plant mask (B) for sim_advance / step: an instance with active[b] == 0 keeps its x0 (a car that failed or finished its lap is
 * frozen, as the reference's loop stops, python/main.py:503-517); NULL = all active */
int ihm2mpc_set_active(ihm2mpc_handle *h, const int32_t *active);
/* closed loops longer than the three laps the track tables hold (python/motion_planning.py:405-430): with enable != 0,
 * prepare_step / step first move every car that has passed s = L (L = -s_ref[0], Track::length) back by one lap -- x0 and the
 * s-component of its iterate.  Off by default: the reference's loop stops one metre after a lap (python/main.py:514-517). */
int ihm2mpc_set_lap_wrap(ihm2mpc_handle *h, int32_t enable);
/* pipelined read-back: u0 (B,2) of the last solve is copied to PINNED host memory (ihm2mpc_host_alloc) in stream order, without
 * waiting -- the next ihm2mpc_step can be enqueued at once; the data is valid after ihm2mpc_synchronize (or any blocking getter) */
int ihm2mpc_host_alloc(uint64_t nbytes, void **p);
int ihm2mpc_host_free(void *p);
int ihm2mpc_get_u0_async(ihm2mpc_handle *h, double *pinned_dst);

/* IHM2Controller.compute_control (python/main.py:297-334) in ONE call for the whole batch: x0 (B,8) in, reference ramp +
 * warm-start shift + solve (one RTI iteration, or the configured SQP iterations), u0 (B,2) and status (B, may be NULL) out;
 * one host-device round trip and one wait -- the call of a real-time controller (mpc_control_node.cpp:105-255). */
int ihm2mpc_compute_control(ihm2mpc_handle *h, const double *x0, double s_target, double *u0, int32_t *status);
/* device room for the histories of up to n_steps steps (run_steps grows it on demand; reserving keeps the allocations out of
 * a timed call) */
int ihm2mpc_reserve_history(ihm2mpc_handle *h, int32_t n_steps);
/* n_steps control steps of the MiL loop (python/main.py:476-517: plant, reference ramp + shift, one RTI iteration) in ONE
 * launch: every instance runs its steps back to back on its own wavefront, so no instance waits for the slowest QP of the
 * batch at every step (throughput follows the mean interior-point iteration count instead of the maximum).  Results are
 * those of n_steps calls of ihm2mpc_step -- bit for bit where both take the single-wave QP kernel and the batch linearisation, i.e.
 * for batches of more than one instance per compute unit; smaller batches take the latency kernels in ihm2mpc_step (four wavefronts
 * per instance in the QP -- the environment variable IHM2MPC_BLOCK_QP=0 turns that off --, results equal to 1e-9); ihm2mpc_step, ihm2mpc_solve and
 * ihm2mpc_compute_control linearise batches of up to 128 intervals (batch <= 3 at N = 40) one sensitivity column per wavefront
 * (the latency path of the single real-time controller), whose records agree with the loop's to 1e-15, not bit for bit.  The persistent loop exists for the three OCP models (fkin6, fdyn6,
 * fdyn6u) -- all-hard constraint tables (the reference's OCP) and soft / track-row tables with batch-shared weights and rows, in the RTI and the SQP mode --
 * and pays off while every instance has a wavefront of its own (batch <= 4 per compute unit); any other case runs
 * n_steps x ihm2mpc_step internally (freeze == 0) or is refused (freeze != 0).
 * freeze != 0: the rules of the reference's loop per car -- a solve status other than 0 / 2 (python/main.py:326-328) or a NaN
 * plant state (:503-504) stops the car where it is, s > lap_stop ends its run (:514-517); the plant mask of
 * ihm2mpc_set_active is updated accordingly and kept across calls until ihm2mpc_set_active(NULL).  freeze == 0: a car masked
 * by ihm2mpc_set_active keeps solving, only its plant stands still (as in ihm2mpc_step).  Histories (any may be NULL): u0 (n_steps,B,2), x0 after the plant
 * (n_steps,B,8), status and QP iterations (n_steps,B); copied in stream order (pinned destinations do not block).  ihm2mpc_get_residuals
 * afterwards: the NLP residuals of the iterate the LAST step started from, as after n_steps calls of ihm2mpc_step (in the RTI mode the
 * loop does not form them on the steps before -- they are an output, not an input of the iteration; a car that stopped earlier in the
 * launch keeps what an earlier call left). */
int ihm2mpc_run_steps(ihm2mpc_handle *h, int32_t model, int32_t M_sim, double s_target, int32_t n_steps, int32_t freeze,
                      double lap_stop, double *u0_hist, double *x0_hist, int32_t *status_hist, int32_t *qp_iter_hist);

/* ---- Cartesian side of the ROS stack ----
 * Plants of the simulation node (src/ihm2/src/sim_node.cpp:197-257), state (X, Y, phi, v_x, v_y, r, T, delta): */
#define IHM2MPC_PLANT_KIN6 3    /* python/models.py:168-229 */
#define IHM2MPC_PLANT_DYN6 4    /* python/models.py:310-452 (implicit there; solved for xdot on device) */
#define IHM2MPC_PLANT_ROS (-3)  /* kin6 while hypot(v_x, v_y) < v_dyn, else dyn6; no reversing (sim_node.cpp:200,246-250) */
/* centre line of every track on the s_ref grid of ihm2mpc_set_tracks: X_ref, Y_ref, phi_ref (ntracks, nknots) -- the columns of
 * the track file the C++ Track loads (src/ihm2/src/common/tracks.cpp:132-181) */
int ihm2mpc_set_track_geometry(ihm2mpc_handle *h, const double *X_ref, const double *Y_ref, const double *phi_ref);
/* n_steps plant steps of length dt_sim (the node: 0.01 s), each RK4 x M_sim, under a constant u; host (B,8),(B,2) -> (B,8) */
int ihm2mpc_sim_step_cart(ihm2mpc_handle *h, int32_t model, int32_t M_sim, double dt_sim, int32_t n_steps, double v_dyn,
                          const double *x, const double *u, double *x_next);
/* plant step of the 15-state Frenet model with wheel speeds, fdyn10 (python/models.py:609-801; the DYN10 plant of python/main.py:490-502):
 * x (B,15) = (s, n, psi, v_x, v_y, r, omega_FL, omega_FR, omega_RL, omega_RR, tau_FL, tau_FR, tau_RL, tau_RR, delta), u (B,5) = (four
 * torque commands, u_delta), over the handle's dt on the handle's track tables, with the handle's plant integrator
 * (ihm2mpc_config.sim_integrator_type):
 *   IHM2MPC_INTEG_IRK_RADAU4 -- replaces AcadosSimSolver.simulate of python/main.py:395-400,490-502 (IRK, GAUSS_RADAU_IIA, 4 stages,
 *     num_steps = M_sim): collocation steps of at most dt / M_sim, each solved to convergence (at most IHM2MPC_DYN10_NEWTON_MAX Newton
 *     iterations, else the step is cut by four), so that the reference's start at REST (python/main.py:438-441) integrates -- at
 *     standstill the slip-ratio denominators smooth_abs_nonzero(0) = 1e-6 defeat a fixed number of Newton iterations; NaN rows if a
 *     step cannot be made;
 *   IHM2MPC_INTEG_ERK -- RK4 x M_sim, for moving cars only. */
#define IHM2MPC_DYN10_NEWTON_MAX 10
int ihm2mpc_sim_step_dyn10(ihm2mpc_handle *h, int32_t M_sim, const double *x, const double *u, double *x_next);
/* Track::project (tracks.cpp:183-288) + the Frenet states of the control node (src/ihm2/src/mpc_control_node.cpp:142-157):
 * x_cart (B,8) -> x_frenet (B,8) = (s, n, psi, v_x, v_y, r, T, delta); s_guess (B) in: centre of the search window of
 * half-width s_tol (the node: 2.0), out: fmod(s + 0.05 v_x, lap length) */
int ihm2mpc_project(ihm2mpc_handle *h, const double *x_cart, double *s_guess, double s_tol, double *x_frenet);
/* device-resident Cartesian plant state for closed loops: sim_advance_cart = plant(x_cart, u0 of the last solve), then
 * x0 <- project(x_cart) */
int ihm2mpc_set_cart_state(ihm2mpc_handle *h, const double *x_cart, const double *s_guess);
int ihm2mpc_get_cart_state(ihm2mpc_handle *h, double *x_cart, double *s_guess);
int ihm2mpc_sim_advance_cart(ihm2mpc_handle *h, int32_t model, int32_t M_sim, double dt_sim, int32_t n_steps, double v_dyn,
                             double s_tol);

/* ---- multi-GPU: the one exchange step of the path, RCCL over xGMI behind this ABI, no PyTorch ----
 * Instances are independent: every GPU owns a contiguous block of the global batch and no collective runs on the data path; the
 * results (u0, status: 20 bytes per instance) are gathered once at the end.  Blocks may differ in size (block split): they are
 * padded to the largest block for the collective and trimmed on the host. */
typedef struct ihm2mpc_group ihm2mpc_group;
/* single process, one handle per device (ncclCommInitAll) */
int ihm2mpc_group_create(ihm2mpc_handle *const *handles, int32_t n, ihm2mpc_group **out);
/* u0_all (sum of the batches, 2), status_all (sum of the batches): host, in handle order */
int ihm2mpc_group_allgather_results(ihm2mpc_group *g, double *u0_all, int32_t *status_all);
int ihm2mpc_group_free(ihm2mpc_group *g);
/* one process per GPU (as `python -m torch.distributed.run` launches bench.py): rank 0 makes the 128-byte id, a side channel
 * carries it to the others (ihm2_amd/dist.py: a TCP socket on MASTER_ADDR), every rank joins; sizes (world) = instances per rank */
int ihm2mpc_comm_unique_id(uint8_t *id128);
int ihm2mpc_comm_init(ihm2mpc_handle *h, int32_t world, int32_t rank, const uint8_t *id128, const int32_t *sizes);
int ihm2mpc_comm_allgather_results(ihm2mpc_handle *h, double *u0_all, int32_t *status_all);      /* host, rank order, on every rank */
int ihm2mpc_comm_allreduce_max(ihm2mpc_handle *h, double *value);      /* in place; doubles as a barrier */
/* what the communicator spans, for the record of a multi-GPU run: *count <- ncclCommCount; device_ids (world) <- the PCI identity
 * (domain << 24 | bus << 8 | device; bits 48..62: a tag of the device's UUID, which tells partitions of one package apart) of every rank's
 * device, gathered over the communicator: N ranks on fewer than N devices repeat one */
int ihm2mpc_comm_info(ihm2mpc_handle *h, int32_t *count, int64_t *device_ids);
int ihm2mpc_comm_free(ihm2mpc_handle *h);

#ifdef __cplusplus
}
#endif
#endif
