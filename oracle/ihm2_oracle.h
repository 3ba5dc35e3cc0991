/*
 * ihm2_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C, double-precision, readable restatement of the reference's NMPC hot path
 * (tudoroancea/ihm2: python/models.py, python/mpc.py, python/main.py) plus the
 * SQP-RTI arithmetic that the reference delegates to acados/HPIPM (third party,
 * unpinned, absent from the reference tree -- see SURVEY.md F1/F2).
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for this
 * path and acados/casadi cannot be imported here, so this oracle is pinned only by
 * (i) the golden constants imported from python/constants.py (tests/golden/constants.json)
 * and (ii) self-certifying checks (complex-step Jacobians, high-accuracy ODE reference,
 * finite-difference sensitivities, KKT residuals, scipy cross-checks) in tests/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
 * library. The product (ihm2_amd/) never links, imports or calls it.
 *
 * Conventions: x = (s, n, psi, v_x, v_y, r, T, delta), u = (u_T, u_delta)
 * (python/models.py:236-245). All arrays are C-contiguous doubles.
 */
#ifndef IHM2_ORACLE_H
#define IHM2_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NX 8
#define ORC_NU 2
#define ORC_NZ 10
#define ORC_NY 12
#define ORC_NYE 8
#define ORC_NG 2
#define ORC_NH 2
/* one-sided inequalities per stage: lower (8 bx, 2 bu, 2 g, 2 h) then upper (8, 2, 2, 2).
 * A second build of the same sources with -DORC_NC=15 (libihm2_oracle_nc15.so, oracle/Makefile) carries the lateral-acceleration row of the
 * kinematic constraint set (old/generate_acaods_interface.py:198-209) as row 14: every (.., ORC_NC, ..) and (.., 2 ORC_NC) array below is then
 * 15 / 30 wide -- lower sides (8, 2, 2, 2, 1), then upper sides */
#ifndef ORC_NC
#define ORC_NC 14
#endif
#define ORC_NMAX 128

/* FDYN6U: fdyn6 with every wheel's lateral force on its OWN slip angle; the reference crosses them (python/models.py:543-546,
 * quirk Q3), which makes the model open-loop unstable (yaw eigenvalue +34 1/s at 10 m/s) */
/* KIN6 / DYN6: the Cartesian plants of the ROS simulation node (python/models.py:168-229, 310-452), state (X, Y, phi, ...) */
enum { ORC_MODEL_FKIN6 = 0, ORC_MODEL_FDYN6 = 1, ORC_MODEL_FDYN6U = 2, ORC_MODEL_KIN6 = 3, ORC_MODEL_DYN6 = 4 };
/* ORC_INTEG_RK4: classical RK4 x M (acados ERK, 4 stages, num_steps = M; old/generate.py:23-25).
 * ORC_INTEG_IRK_GL4 / _RADAU4: 4-stage collocation, M steps, 3 Newton iterations per step from K = 0 with a fresh Jacobian each (acados
 * IRK defaults: sim_method_newton_iter 3, jac_reuse 0) -- python/main.py:234-236 (OCP: Gauss-Legendre, 1 step),
 * python/main.py:395-400 (plants: Radau IIA, 100 steps); forward sensitivities by the implicit-function theorem at the final stage values */
enum { ORC_INTEG_RK4 = 0, ORC_INTEG_IRK_GL4 = 1, ORC_INTEG_IRK_RADAU4 = 2 };
#define ORC_IRK_NEWTON_ITER 3
#define ORC_IPM_STEP_FRACTION 0.995 /* fraction of the distance to the boundary an interior-point step may take: the value of IHM2MPC_IPM_STEP_FRACTION (tests/test_cabi.py compares them) */

typedef struct {
    int N;       /* shooting intervals */
    int M;       /* RK4 sub-steps per interval */
    int model;   /* ORC_MODEL_* */
    int integrator; /* ORC_INTEG_* */
    double dt;   /* interval length */
    double cost_scale_stage; /* factor on stage cost terms k<N (acados: time step) */
    /* track tables: ntracks x (s_ref[nknots], kappa_ref[nknots]) */
    int ntracks;
    int nknots;
    const double *s_ref;     /* (ntracks, nknots) */
    const double *kappa_ref; /* (ntracks, nknots) */
    /* cost, shared by the batch */
    const double *W;   /* (N, 12, 12) */
    const double *W_e; /* (8, 8) */
    /* bounds, shared by the batch; +-inf (|v| >= 1e20) = absent */
    const double *lbx; /* (N+1, 8) ; stage 0 row ignored (x_0 is fixed to x0) */
    const double *ubx; /* (N+1, 8) */
    const double *lbu; /* (N, 2) */
    const double *ubu; /* (N, 2) */
    const double *C;   /* (N, 2, 8) */
    const double *D;   /* (N, 2, 2) */
    const double *lg;  /* (N, 2) */
    const double *ug;  /* (N, 2) */
    /* soft constraint sides (old/generate_acaods_interface.py:380-395): (N+1,28) per one-sided constraint,
     * 14 lower then 14 upper; soft_Z < 0 = hard; NULL = all hard */
    const double *soft_z;
    const double *soft_Z;
    /* nonlinear track-boundary rows (old/generate_acaods_interface.py:191-212), stages 1..N:
     *   h_R = n - 1/2 L sin|psi| + 1/2 W cos(psi) - w_R ,  h_L = -n + 1/2 L sin|psi| + 1/2 W cos(psi) - w_L
     * with lh <= h <= uh (|bound| >= 1e20 = absent); widths (ntracks,2) = (w_R, w_L), constant along a track as in
     * python/motion_planning.py:385-386; path_on = 0 disables the rows */
    int path_on;
    double car_L, car_W;
    const double *widths;
    double lh[ORC_NH], uh[ORC_NH];
    /* interior-point options */
    int ipm_iter_max;
    double ipm_tol;  /* abs inf-norm tolerance on all four residual groups */
    double ipm_mu0;
    double ipm_tau0; /* lower clamp for initial slacks */
    /* lateral-acceleration row (ORC_NC = 15 builds only; ignored otherwise), stages 1..N-1 of the kinematic model:
     *   alat_lb <= a_lat(x_k) <= alat_ub ,  a_lat = (-F_Rx sin(beta) + F_Fx sin(delta - beta)) / m + (v_x^2 + v_y^2) sin(beta) / l_R
     * old/generate_acaods_interface.py:198-209 (row), :266-271 and old/scripts/gen_mpc.py:182-184 (definition), :52-53 (bounds -5 / +5) */
    int alat_on;
    double alat_lb, alat_ub;
} orc_problem;

/* --- model (python/models.py:232-307 fkin6, :455-606 fdyn6) --- */
double orc_kappa(const double *s_ref, const double *kappa_ref, int nknots, double s, double *dkappa_ds);
void orc_f(int model, const double *x, const double *u, const double *s_ref, const double *kappa_ref,
           int nknots, double *xdot);
/* J is 8x10 row-major: d xdot / d (x, u) */
void orc_jac(int model, const double *x, const double *u, const double *s_ref, const double *kappa_ref,
             int nknots, double *xdot, double *J);
/* same Jacobian by complex-step evaluation of the model formulas (both models) */
void orc_jac_cs(int model, const double *x, const double *u, const double *s_ref, const double *kappa_ref,
                int nknots, double *xdot, double *J);

/* lateral acceleration of the kinematic model at x (8) and its gradient with respect to x (8; non-zeros in v_x, v_y, T, delta), by complex-step
 * evaluation of the formula as the reference writes it (old/scripts/gen_mpc.py:182-184 with the forces of python/models.py:255-258) */
double orc_alat(const double *x, double *grad);

/* --- integrator: RK4 x M with forward sensitivities; A 8x8, Bm 8x2 row-major --- */
void orc_rk4_sens(int model, int integrator, const double *x, const double *u, const double *s_ref,
                  const double *kappa_ref, int nknots, double dt, int M, double *xnext, double *A,
                  double *Bm);
void orc_rk4(int model, int integrator, const double *x, const double *u, const double *s_ref,
             const double *kappa_ref, int nknots, double dt, int M, double *xnext);

/* --- stage-wise QP by Riccati-based primal-dual interior point ---
 * H (N+1,10,10), g (N+1,10), A (N,8,8), Bm (N,8,2), b (N,8), dx0 (8),
 * R (N+1, 14, 10) constraint rows, dl/du (N+1, 14) (+-inf = absent)
 * out: dz (N+1,10), pi (N+1,8), lam (N+1,28) [14 lower then 14 upper], t (N+1,28)
 * stats[0..3] = res_g,res_b,res_d,res_m at exit, stats[4]=mu, stats[5..6] = scales sg, sb (stats has 8 slots)
 * tolerances are relative: res_g,res_m <= tol*sg, res_b,res_d <= tol*sb; mu0 is a factor on sg.
 * returns 0 converged, 1 max-iter but within 1e4*tol, 2 min step, 3 NaN, 4 max-iter and not converged */
int orc_qp_solve(int N, const double *H, const double *g, const double *A, const double *Bm,
                 const double *b, const double *dx0, const double *R, const double *dl,
                 const double *du, int iter_max, double tol, double mu0, double tau0, double *dz,
                 double *pi, double *lam, double *t, double *stats, int *iters);

/* same with SOFT constraint sides: soft_z, soft_Z (N+1,28) per one-sided constraint (12 lower then 12 upper);
 * a side with soft_Z >= 0 carries a slack s >= 0 with cost soft_z s + 1/2 soft_Z s^2; sl (N+1,28): slack values out */
int orc_qp_solve_soft(int N, const double *H, const double *g, const double *A, const double *Bm,
                      const double *b, const double *dx0, const double *R, const double *dl,
                      const double *du, const double *soft_z, const double *soft_Z, int iter_max, double tol,
                      double mu0, double tau0, double *dz, double *pi, double *lam, double *t, double *sl,
                      double *stats, int *iters);

/* --- one SQP-RTI iteration for a batch (OpenMP over instances) ---
 * x (B,N+1,8), u (B,N,2), x0 (B,8), yref (B,N,12), yref_e (B,8), track_id (B)
 * pi (B,N+1,8), lam (B,N+1,24): multipliers, in/out
 * status (B) int32: 0 ok, 1 NaN, 4 QP failure (acados codes, dpc/main.py:287-293); a failed
 * instance keeps its iterate and multipliers unchanged
 * res (B,4): NLP KKT residual inf-norms (stat, eq, ineq, comp) at the iterate BEFORE the step
 * qp_iter (B) */
void orc_rti_step(const orc_problem *P, int B, double *x, double *u, const double *x0,
                  const double *yref, const double *yref_e, const int *track_id, double *pi,
                  double *lam, int *status, double *res, int *qp_iter, int nthreads);

/* --- globalised SQP (python/main.py:230-237: "SQP", max_iter 2, "MERIT_BACKTRACKING"): up to max_iter iterations of
 * [KKT test at the iterate -> QP -> backtracking line search on the l1 merit function -> step alpha on primal and dual];
 * see ihm2_oracle_rti.c.  globalization 0 = FIXED_STEP (alpha = 1).  sl (B,N+1,28): slack values, in/out (start at 0);
 * status: 0 converged, 2 max_iter, 4 QP failure, 1 NaN; sqp_iter (B): QP solves made; alpha (B): last step length */
typedef struct {
    int max_iter;
    int globalization;
    int use_sufficient_descent;
    int full_step_dual;
    double tol[4];      /* stat, eq, ineq, comp */
    double alpha_min, alpha_reduction, eps_sufficient_descent;
} orc_sqp_opts;
void orc_sqp_solve(const orc_problem *P, const orc_sqp_opts *O, int B, double *x, double *u, const double *x0,
                   const double *yref, const double *yref_e, const int *track_id, double *pi, double *lam, double *sl,
                   int *status, double *res, int *qp_iter, int *sqp_iter, double *alpha, int nthreads);

/* linearisation only: A (B,N,8,8), Bm (B,N,8,2), b (B,N,8) where b = Phi(x_k,u_k) - x_{k+1} */
void orc_linearize(const orc_problem *P, int B, const double *x, const double *u,
                   const int *track_id, double *A, double *Bm, double *b, int nthreads);

/* QP data of one instance as the RTI step assembles it (for QP-level parity tests) */
void orc_build_qp(const orc_problem *P, const double *x, const double *u, const double *x0,
                  const double *yref, const double *yref_e, int track_id, double *H, double *g,
                  double *A, double *Bm, double *b, double *dx0, double *R, double *dl, double *du);

/* warm-start shift + reference ramp of IHM2Controller.compute_control (python/main.py:297-322) */
void orc_prepare_step(int N, int B, const double *x0, double s_target, double *x, double *u,
                      double *yref, double *yref_e);

/* plant step for closed-loop (python/main.py:476-502): RK4 x M on the chosen model */
/* plant step with a chosen integrator (ORC_INTEG_*) */
void orc_sim_step_integ(const orc_problem *P, int B, int model, int integrator, int M, const double *x, const double *u,
                        const int *track_id, double *xnext, int nthreads);
void orc_sim_step(const orc_problem *P, int B, int model, int M, const double *x, const double *u,
                  const int *track_id, double *xnext, int nthreads);

int orc_num_threads(void);

/* --- Cartesian side of the ROS stack (ihm2_oracle_track.c): projection on the track, Frenet states, plant step --- */
void orc_project(const double *s_ref, const double *X_ref, const double *Y_ref, const double *phi_ref, int nk, double X,
                 double Y, double s_guess, double s_tol, double *s_proj, double *X_proj, double *Y_proj, double *phi_proj);
void orc_cart_to_frenet(int B, int ntracks, int nk, const double *s_ref, const double *X_ref, const double *Y_ref,
                        const double *phi_ref, const int *track_id, const double *x_cart, double *s_guess, double s_tol,
                        double *x_frenet);
void orc_sim_step_cart(int B, int model, int M, double dt, double v_dyn, const double *x, const double *u, double *xnext);

/* --- fdyn10 (ihm2_oracle_dyn10.c): the 15-state Frenet plant with wheel speeds, python/models.py:609-801 --- */
void orc_f_dyn10(const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots, double *xdot);
void orc_sim_step_dyn10(int B, int M, double dt, const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots,
                        double *xnext);
/* f (15) and d f / d x (15 x 15, row-major) by dual-number evaluation of the same formulas */
void orc_jac_dyn10(const double *x, const double *u, const double *s_ref, const double *kappa_ref, int nknots, double *f, double *J);
/* the reference's plant integrator (python/main.py:395-400: IRK, 4 Radau IIA stages, 100 steps over dt): collocation steps of at most
 * dt / M, each solved to convergence (at most newton_iter Newton iterations with fresh Jacobians, else the step is cut by four);
 * usable from rest (python/main.py:438-441); NaN if a step cannot be made */
void orc_sim_step_dyn10_irk(int B, int integrator, int M, int newton_iter, double dt, const double *x, const double *u, const double *s_ref,
                            const double *kappa_ref, int nknots, double *xnext);

#ifdef __cplusplus
}
#endif
#endif
