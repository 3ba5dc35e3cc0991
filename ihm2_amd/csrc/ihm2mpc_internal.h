// ihm2mpc_internal.h -- handle layout and kernel launchers shared by the .hip translation units.
//
// Device data layout (DESIGN.md "Data layout in HBM"): structure-of-arrays, instance-minor.
// Element `e` of a per-instance array lives at  base[e * Bp + b]  where b is the instance and Bp the
// batch padded to a multiple of 64, so that the 64 lanes of a wavefront working on 64 consecutive
// instances issue one coalesced 512-byte access per element.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ihm2mpc.h"

#define NX 8
#define NU 2
#define NZ 10
#define NY 12
#define NG 2
#define NC 12   // two-sided constraints per stage: 8 state boxes, 2 input boxes, 2 general rows
#define NLAM 24

struct ihm2mpc_handle {
    ihm2mpc_config cfg;
    int B, Bp, N, NS;
    hipStream_t stream;
    hipEvent_t ev[4];
    float ms_total, ms_lin, ms_qp;
    bool tracks_set, weights_set, bounds_set;

    // ---- shared problem data (device) ----
    double *s_ref, *kappa_ref;     // (ntracks, nknots)
    int32_t *track_id;             // (Bp)
    double *Hs;                    // (NS,10,10)  cost_scale * V'WV ; terminal: W_e padded
    double *Gy;                    // (NS,10,12)  cost_scale * V'W  ; terminal: W_e in the first 8x8
    double *lbx, *ubx;             // (NS,8)
    double *lbu, *ubu;             // (N,2)
    double *CD;                    // (N,2,10)  general rows [C D]
    double *lg, *ug;               // (N,2)

    // ---- per-instance state, SoA [elem][Bp] ----
    double *x;      // (NS*8)
    double *u;      // (N*2)
    double *x0;     // (8)
    double *yref;   // (N*12)
    double *yref_e; // (8)
    double *pi;     // (NS*8)
    double *lam;    // (NS*24)
    double *res;    // (4)
    int32_t *status, *qp_iter;   // (Bp)
    double *u0;     // (2) control of the last solve

    // ---- linearisation ----
    double *A;      // (N*64)
    double *Bm;     // (N*16)
    double *bvec;   // (N*8)

    // ---- QP workspace (per instance, SoA) ----
    double *q_g;    // (NS*10) gradient
    double *q_dl, *q_du;   // (NS*12)
    double *q_z;    // (NS*10)
    double *q_pi;   // (NS*8)
    double *q_lam, *q_t;   // (NS*24)
    double *q_gt;   // (NS*10)
    double *q_rb;   // (N*8)
    double *q_rd;   // (NS*24)
    double *q_dz;   // (NS*10)
    double *q_dpi;  // (NS*8)
    double *q_dlam, *q_dt, *q_dlam_a, *q_dt_a; // (NS*24)
    double *q_P;    // (NS*36) packed symmetric
    double *q_Gux;  // (N*16)
    double *q_Ginv; // (N*3)
    double *q_p;    // (NS*8)
    double *q_kff;  // (N*2)

    // ---- staging ----
    double *stage_d;   // device AoS staging, max(NS*24, ...)*B doubles
    double *stage_h;   // pinned host staging of the same size
    size_t stage_elems;
};

// --- launchers (each defined in one .hip file) ---
void ihm2_launch_aos_to_soa(ihm2mpc_handle *h, const double *aos, double *soa, int elems);
void ihm2_launch_soa_to_aos(ihm2mpc_handle *h, const double *soa, double *aos, int elems);
void ihm2_launch_fill(ihm2mpc_handle *h, double *soa, int elems, double value);
void ihm2_launch_prepare(ihm2mpc_handle *h, double s_target);
void ihm2_launch_init_guess(ihm2mpc_handle *h, double v_ref_scale);
void ihm2_launch_linearize(ihm2mpc_handle *h);
void ihm2_launch_qp(ihm2mpc_handle *h);
void ihm2_launch_sim(ihm2mpc_handle *h, int model, int M_sim, const double *x_soa, const double *u_soa, double *xn_soa);
