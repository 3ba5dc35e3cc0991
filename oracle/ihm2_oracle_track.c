/*
 * ihm2_oracle_track.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Cartesian -> Frenet projection of the ROS stack, restated from
 *   Track::project                src/ihm2/src/common/tracks.cpp:183-288   (windowed nearest point, segment choice by the
 *                                                                            angle seen from the car, projection on the segment,
 *                                                                            phi_ref from the linear spline OF THE NEAREST POINT)
 *   MPCControlNode, Frenet states src/ihm2/src/mpc_control_node.cpp:142-157 (psi = wrap(phi - phi_ref), n = +-distance with the
 *                                                                            sign of tangent x offset, s_guess update)
 *   wrap_to_pi                    src/ihm2/src/common/math.cpp:19-25
 * and the plant step of the simulation node (kin6 / dyn6 speed switch, no reversing)
 *   SimNode::sim_timer_cb         src/ihm2/src/sim_node.cpp:197-257.
 */
#include <math.h>
#include <stddef.h>

#include "ihm2_oracle.h"

static double wrap_to_pi(double x)
{
    double t = fmod(x + M_PI, 2.0 * M_PI);
    if (t < 0) t += 2.0 * M_PI;
    return t - M_PI;
}

/* index of the last element <= x (std::upper_bound - 1), -1 if x < v[0] */
static int locate_index(const double *v, int n, double x)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (v[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

static double angle3pt(double ax, double ay, double bx, double by, double cx, double cy)
{
    return wrap_to_pi(atan2(cy - by, cx - bx) - atan2(ay - by, ax - bx));
}

void orc_project(const double *s_ref, const double *X_ref, const double *Y_ref, const double *phi_ref, int nk, double X,
                 double Y, double s_guess, double s_tol, double *s_proj, double *X_proj, double *Y_proj, double *phi_proj)
{
    const double s_low = fmax(s_guess - s_tol, s_ref[0]), s_up = fmin(s_guess + s_tol, s_ref[nk - 1]);
    int id_low = locate_index(s_ref, nk, s_low), id_up = locate_index(s_ref, nk, s_up);
    if (id_low > 0) --id_low;
    if (id_up < nk - 1) ++id_up;
    const int nloc = id_up - id_low + 1;
    int id_min = 0;
    double best = INFINITY;
    for (int i = 0; i < nloc; i++) {
        const double dx = X_ref[id_low + i] - X, dy = Y_ref[id_low + i] - Y, d = dx * dx + dy * dy;
        if (d < best) { best = d; id_min = i; }
    }
    const int id_prev = (id_min == 0) ? nloc - 1 : id_min - 1;          /* wraps inside the window, as written */
    const int id_next = (id_min == nloc - 1) ? 0 : id_min + 1;
    const double mx = X_ref[id_low + id_min], my = Y_ref[id_low + id_min];
    const double px = X_ref[id_low + id_prev], py = Y_ref[id_low + id_prev];
    const double nx = X_ref[id_low + id_next], ny = Y_ref[id_low + id_next];
    const double angle_prev = fabs(angle3pt(mx, my, X, Y, px, py)), angle_next = fabs(angle3pt(mx, my, X, Y, nx, ny));
    double ax, ay, bx, by, sa, sb;
    if (angle_prev > angle_next) { ax = px; ay = py; bx = mx; by = my; sa = s_ref[id_prev + id_low]; sb = s_ref[id_min + id_low]; }
    else { ax = mx; ay = my; bx = nx; by = ny; sa = s_ref[id_min + id_low]; sb = s_ref[id_next + id_low]; }
    const double dx = bx - ax, dy = by - ay;
    const double lambda = ((X - ax) * dx + (Y - ay) * dy) / (dx * dx + dy * dy);
    *s_proj = sa + lambda * (sb - sa);
    *X_proj = ax + lambda * (bx - ax);
    *Y_proj = ay + lambda * (by - ay);
    /* Track::interp with ind = id_min + id_low: the linear piece that STARTS at the nearest point (clamped to the last piece) */
    int ind = id_min + id_low;
    if (ind > nk - 2) ind = nk - 2;
    *phi_proj = phi_ref[ind] + (phi_ref[ind + 1] - phi_ref[ind]) / (s_ref[ind + 1] - s_ref[ind]) * (*s_proj - s_ref[ind]);
}

/* x_cart (B,8) = (X, Y, phi, v_x, v_y, r, T, delta) -> x_frenet (B,8) = (s, n, psi, v_x, ...); s_guess (B) is updated to
 * fmod(s + 0.05 v_x, lap length) with lap length = -s_ref[0] (Track::length, tracks.cpp:308) */
void orc_cart_to_frenet(int B, int ntracks, int nk, const double *s_ref, const double *X_ref, const double *Y_ref,
                        const double *phi_ref, const int *track_id, const double *x_cart, double *s_guess, double s_tol,
                        double *x_frenet)
{
    (void)ntracks;
    for (int b = 0; b < B; b++) {
        const int t = track_id ? track_id[b] : 0;
        const double *sr = s_ref + (size_t)t * nk, *Xr = X_ref + (size_t)t * nk, *Yr = Y_ref + (size_t)t * nk, *pr = phi_ref + (size_t)t * nk;
        const double *xc = x_cart + (size_t)b * 8;
        double s, Xp, Yp, phi_p;
        orc_project(sr, Xr, Yr, pr, nk, xc[0], xc[1], s_guess[b], s_tol, &s, &Xp, &Yp, &phi_p);
        const double rho = wrap_to_pi(phi_p);
        const double psi = wrap_to_pi(wrap_to_pi(xc[2]) - rho);
        const double e = hypot(Xp - xc[0], Yp - xc[1]);
        const double tpr = (xc[1] - Yp) * cos(rho) - (xc[0] - Xp) * sin(rho);
        double *xf = x_frenet + (size_t)b * 8;
        xf[0] = s;
        xf[1] = e * (tpr > 0.0 ? 1.0 : -1.0);      /* the node throws when |tpr| < 1e-6; a batch keeps the (then tiny) offset */
        xf[2] = psi;
        for (int i = 3; i < 8; i++) xf[i] = xc[i];
        s_guess[b] = fmod(s + xc[3] * 0.05, -sr[0]);
    }
}

/* one plant step of the simulation node: kin6 below v_dyn, dyn6 above (speed of the state BEFORE the step), RK4 x M over dt,
 * then "prohibit the car from going backwards" (sim_node.cpp:246-250).  model: ORC_MODEL_KIN6, ORC_MODEL_DYN6 or -3 = switch */
void orc_sim_step_cart(int B, int model, int M, double dt, double v_dyn, const double *x, const double *u, double *xnext)
{
    for (int b = 0; b < B; b++) {
        const double *xb = x + (size_t)b * 8;
        int mdl = model;
        if (model == -3) mdl = (hypot(xb[3], xb[4]) < v_dyn) ? ORC_MODEL_KIN6 : ORC_MODEL_DYN6;
        double *xn = xnext + (size_t)b * 8;
        orc_rk4(mdl, ORC_INTEG_RK4, xb, u + (size_t)b * 2, 0, 0, 0, dt, M, xn);
        if (model == -3 && (xn[3] < 0.0 || (xn[6] <= 0.1 && xn[3] < 0.01))) { xn[3] = 0.0; xn[4] = 0.0; xn[5] = 0.0; }
    }
}
