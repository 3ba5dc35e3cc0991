// api.hip -- the C ABI of libihm2mpc.so (include/ihm2mpc.h): handle lifetime, copy-in setters,
// copy-out getters, and the launch sequence of one RTI iteration on the handle's HIP stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ihm2mpc_internal.h"

static thread_local std::string g_err;

// the error string of ihm2mpc_last_error, also set by comm.hip
int ihm2_fail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

namespace {

#define fail(...) ihm2_fail(__VA_ARGS__)

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define CHECK_H(h)                                        \
    do {                                                  \
        if (!(h)) return fail("null handle");             \
        HIP_TRY(hipSetDevice((h)->cfg.device));           \
    } while (0)

template <typename T>
int dalloc(T **p, size_t n)
{
    HIP_TRY(hipMalloc((void **)p, n * sizeof(T)));
    HIP_TRY(hipMemset(*p, 0, n * sizeof(T)));
    // the handle's streams are non-blocking: they do not wait for the null stream the fill runs on, and an upload that
    // overtakes it would be zeroed afterwards
    HIP_TRY(hipStreamSynchronize(nullptr));
    return 0;
}

// host (B, elems) <-> device (B, elems): same instance-major layout on both sides
int upload(ihm2mpc_handle *h, const double *host, double *dev, int elems)
{
    HIP_TRY(hipMemcpyAsync(dev, host, (size_t)h->B * elems * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // the caller's buffer may be reused as soon as we return
    return 0;
}

int download(ihm2mpc_handle *h, const double *dev, double *host, int elems)
{
    HIP_TRY(hipMemcpyAsync(host, dev, (size_t)h->B * elems * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int upload_shared(ihm2mpc_handle *h, const double *host, double *dev, size_t n)
{
    HIP_TRY(hipMemcpyAsync(dev, host, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// y = Vx x + Vu u of python/mpc.py:49-58 as one 12x10 selector
void cost_selector(double V[NY][NZ])
{
    memset(V, 0, sizeof(double) * NY * NZ);
    for (int i = 0; i < NX; i++) V[i][i] = 1.0;
    V[10][6] = 1.0; V[11][7] = 1.0;
    V[8][8] = 1.0; V[9][9] = 1.0;
    V[10][8] = -1.0; V[11][9] = -1.0;
}

int ready(ihm2mpc_handle *h)
{
    if (!h->tracks_set) return fail("ihm2mpc_set_tracks has not been called");
    if (!h->weights_set) return fail("ihm2mpc_set_weights has not been called");
    if (!h->bounds_set) return fail("ihm2mpc_set_bounds has not been called");
    if (!h->slots_fit) return fail("the constraint rows fit no QP kernel: at most 10 slots per lane of 64 (a two-sided hard row is one slot, a row with a soft side two), of which at most 4 soft; with soft sides the hard two-sided rows get 10 - 4 or 8 - 3 (8 - 2 without track rows) of them");
    if (h->alat_on) {
        // the row belongs to the kinematic constraint set of old/generate_acaods_interface.py:198-209, which comes with the track rows and SQP_RTI (old/generate.py:21)
        if (h->cfg.model != IHM2MPC_MODEL_FKIN6) return fail("the lateral-acceleration row is a row of the kinematic model (old/generate_acaods_interface.py:206: `[] if is_dynamic else [a_lat]`)");
        if (!h->path_on) return fail("the lateral-acceleration row comes with the track rows: enable ihm2mpc_set_path_constraints first");
        if (h->cfg.nlp_solver_type != IHM2MPC_SQP_RTI) return fail("the lateral-acceleration row is implemented for SQP_RTI (old/generate.py:21)");
        if (!(h->uniform_H && h->uniform_CD)) return fail("the lateral-acceleration row needs stage-independent weights and general rows (the reference's OCP has them)");
    }
    return 0;
}

struct FieldInfo { double *base; int per_stage; int nstages; };

int field_info(ihm2mpc_handle *h, const char *field, FieldInfo *fi)
{
    const std::string f(field ? field : "");
    if (f == "x") *fi = {h->x, NX, h->NS};
    else if (f == "u") *fi = {h->u, NU, h->N};
    else if (f == "yref") *fi = {h->yref, NY, h->N};
    else if (f == "yref_e") *fi = {h->yref_e, NX, 1};
    else if (f == "pi") *fi = {h->pi, NX, h->NS};
    else if (f == "lam") *fi = {h->lam, NLAM, h->NS};
    else if (f == "lbx" || f == "ubx" || f == "x0") *fi = {h->x0, NX, 1};
    else return fail("unknown field '%s'", f.c_str());
    return 0;
}

}  // namespace

// classical RK4 is stable for |z| < 2.785 on the negative real axis: z = -(dt / M) / t for the first-order actuator lags
static bool rk4_unstable(double dt, int M) { return dt / M / 1e-3 >= 2.78; }

extern "C" {

const char *ihm2mpc_last_error(void) { return g_err.c_str(); }
const char *ihm2mpc_version(void) { return "ihm2mpc 0.1 (gfx950)"; }

int ihm2mpc_create(const ihm2mpc_config *cfg, ihm2mpc_handle **out)
{
    if (!cfg || !out) return fail("null argument");
    if (cfg->batch < 1) return fail("batch must be >= 1");
    if (cfg->N < 2 || cfg->N > IHM2MPC_NMAX) return fail("N must be in [2, %d]", IHM2MPC_NMAX);
    if (cfg->M < 1) return fail("M must be >= 1");
    if (cfg->integrator_type < IHM2MPC_INTEG_ERK || cfg->integrator_type > IHM2MPC_INTEG_IRK_RADAU4 || cfg->sim_integrator_type < IHM2MPC_INTEG_ERK ||
        cfg->sim_integrator_type > IHM2MPC_INTEG_IRK_RADAU4)
        return fail("unknown integrator type (%d, %d)", cfg->integrator_type, cfg->sim_integrator_type);
    if (cfg->integrator_type != IHM2MPC_INTEG_ERK && cfg->M != 1)
        return fail("the IRK integrator of the shooting intervals takes one step per interval (sim_method_num_steps = 1, python/main.py:236); M = %d", cfg->M);
    if (cfg->integrator_type == IHM2MPC_INTEG_ERK && rk4_unstable(cfg->dt, cfg->M))
        return fail("RK4 with %d sub-step(s) of dt = %g is unstable on the actuator lags (t_T = 1e-3 s, t_delta = 0.02 s: |z| = dt / (M t) must stay "
                    "below 2.78): use M >= %d (the reference's sim_method_num_steps = 1 belongs to its IRK integrator, python/main.py:234-236)",
                    cfg->M, cfg->dt, (int)ceil(cfg->dt / (2.78 * 1e-3)));
    if (cfg->model < IHM2MPC_MODEL_FKIN6 || cfg->model > IHM2MPC_MODEL_FDYN6U) return fail("unknown OCP model %d", cfg->model);
    if (cfg->ntracks < 1 || cfg->nknots < 2) return fail("need at least one track table with >= 2 knots");
    if (!(cfg->dt > 0.0)) return fail("dt must be positive");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail("device %d out of range (%d devices)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("libihm2mpc is built for gfx950 (MI355X) only; device %d is %s", cfg->device, prop.gcnArchName);

    ihm2mpc_handle *h = new ihm2mpc_handle();
    memset(h, 0, sizeof *h);
    h->cfg = *cfg;
    h->B = cfg->batch;
    h->N = cfg->N;
    h->NS = cfg->N + 1;
    h->n_cu = prop.multiProcessorCount;
    { const char *e = getenv("IHM2MPC_BLOCK_QP"); h->block_qp = !(e && e[0] == '0'); }
    const size_t B = h->B, N = h->N, NS = h->NS;
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&h->ev[i]));
#define DA(p, n) if (dalloc(&h->p, (n))) return -1
    DA(s_ref, (size_t)cfg->ntracks * cfg->nknots); DA(kappa_ref, (size_t)cfg->ntracks * cfg->nknots);
    DA(track_id, B);
    DA(Hs, NS * 100); DA(Gy, NS * 120); DA(lbx, NS * 8); DA(ubx, NS * 8); DA(lbu, N * 2); DA(ubu, N * 2);
    DA(CD, N * 20); DA(lg, N * 2); DA(ug, N * 2);
    DA(slot_kc_blk, 1024); DA(slot_lb_blk, 1024); DA(slot_ub_blk, 1024);
    DA(slot_kc, MAX_SLOTS); DA(slot_lb, MAX_SLOTS); DA(slot_ub, MAX_SLOTS); DA(slot_zw, MAX_SLOTS); DA(slot_Zw, MAX_SLOTS);
    DA(slk, B * NS * NLAM); DA(widths, (size_t)cfg->ntracks * 2); DA(lam_a, B * NS * 2); DA(slk_a, B * NS * 2);
    h->alat_on = 0; h->slots_fit = true; h->alat_lb = -INFINITY; h->alat_ub = INFINITY; h->alat_sz[0] = h->alat_sz[1] = 0.0; h->alat_sZ[0] = h->alat_sZ[1] = -1.0;
    DA(X_ref, (size_t)cfg->ntracks * cfg->nknots); DA(Y_ref, (size_t)cfg->ntracks * cfg->nknots); DA(phi_ref, (size_t)cfg->ntracks * cfg->nknots);
    DA(xc, B * 8); DA(s_guess, B);
    DA(step_args, 32);
    for (int i = 0; i < 2; i++) {      // pinned staging of the persistent loop's argument blocks (uploaded without a host wait)
        HIP_TRY(hipHostMalloc(&h->args_host[i], 1024, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&h->args_ev[i], hipEventDisableTiming));
    }
    DA(Wd, N * 144 + 64); DA(st_lb, NS * NC); DA(st_ub, NS * NC); DA(st_sz, NS * NLAM); DA(st_sZ, NS * NLAM);
    h->sqp_globalization = 0; h->sqp_use_suff = 0; h->sqp_full_step_dual = 0;
    h->sqp_alpha_min = 0.05; h->sqp_alpha_red = 0.7; h->sqp_eps = 1e-4;
    for (int i = 0; i < 4; i++) h->sqp_tol[i] = cfg->nlp_tol;
    h->host_lb = new double[NS * NC]; h->host_ub = new double[NS * NC];
    h->host_sz = new double[NS * NLAM]; h->host_sZ = new double[NS * NLAM];
    for (size_t i = 0; i < NS * NC; i++) { h->host_lb[i] = -INFINITY; h->host_ub[i] = INFINITY; }
    for (size_t i = 0; i < NS * NLAM; i++) { h->host_sz[i] = 0.0; h->host_sZ[i] = -1.0; }
    DA(x, B * NS * 8); DA(u, B * N * 2); DA(x0, B * 8); DA(yref, B * N * 12); DA(yref_e, B * 8);
    DA(pi, B * NS * 8); DA(lam, B * NS * NLAM); DA(res, B * 4); DA(qp_res, B * 4); DA(status, B); DA(qp_iter, B); DA(active, B); DA(u0, B * 2);
    DA(lin, (B * N + B) * LIN_REC);      // + one spare record per instance (the kinematic plant's, never read)
    DA(q_g, B * NS * 10); DA(q_rg, B * NS * 10); DA(q_P, B * NS * 64); DA(q_M, (B * N + 2 * QM_PAD) * 64); DA(scratch, B * 24);
#undef DA
    if (ihm2_upload_irk_tab(h)) return fail("could not upload the collocation tableau");
    *out = h;
    return 0;
}

int ihm2mpc_free(ihm2mpc_handle *h)
{
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    (void)ihm2mpc_comm_free(h);
    void *ptrs[] = {h->s_ref, h->kappa_ref, h->track_id, h->Hs, h->Gy, h->lbx, h->ubx, h->lbu, h->ubu, h->CD, h->lg, h->ug,
                    h->slot_kc, h->slot_lb, h->slot_ub, h->slot_zw, h->slot_Zw, h->slk, h->lam_a, h->slk_a, h->widths, h->X_ref, h->Y_ref, h->phi_ref, h->xc, h->s_guess, h->x, h->u, h->x0, h->yref, h->yref_e, h->pi, h->lam, h->res, h->qp_res, h->dyn10, h->ls_phi, h->slot_kc_blk, h->slot_lb_blk, h->slot_ub_blk,
                    h->status, h->qp_iter, h->active, h->u0, h->lin, h->q_g, h->q_rg, h->q_P, h->q_M, h->scratch, h->step_args, h->Wd, h->st_lb, h->st_ub, h->st_sz, h->st_sZ,
                    h->ls_x, h->ls_u, h->ls_pi, h->ls_lam, h->ls_slk, h->ls_wpi, h->ls_wlam, h->ls_alpha, h->ls_args, h->ls_done, h->ls_status, h->ls_iter, h->ls_qp_acc, h->ls_pending, h->irk_tab, h->sim_irk_tab,
                    h->hist_u0, h->hist_x0, h->hist_st, h->hist_it};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete[] h->host_lb; delete[] h->host_ub; delete[] h->host_sz; delete[] h->host_sZ;
    for (int i = 0; i < 4; i++) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    for (int i = 0; i < 2; i++) { if (h->args_host[i]) (void)hipHostFree(h->args_host[i]); if (h->args_ev[i]) (void)hipEventDestroy(h->args_ev[i]); }
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int ihm2mpc_synchronize(ihm2mpc_handle *h)
{
    CHECK_H(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int ihm2mpc_get_stream(ihm2mpc_handle *h, void **stream)
{
    CHECK_H(h);
    if (!stream) return fail("null argument");
    *stream = (void *)h->stream;
    return 0;
}

int ihm2mpc_set_tracks(ihm2mpc_handle *h, const double *s_ref, const double *kappa_ref)
{
    CHECK_H(h);
    if (!s_ref || !kappa_ref) return fail("null argument");
    const size_t n = (size_t)h->cfg.ntracks * h->cfg.nknots;
    for (int t = 0; t < h->cfg.ntracks; t++)
        for (int i = 1; i < h->cfg.nknots; i++)
            if (!(s_ref[(size_t)t * h->cfg.nknots + i] > s_ref[(size_t)t * h->cfg.nknots + i - 1]))
                return fail("s_ref of track %d is not strictly increasing at knot %d", t, i);
    if (upload_shared(h, s_ref, h->s_ref, n) || upload_shared(h, kappa_ref, h->kappa_ref, n)) return -1;
    h->tracks_set = true;
    return 0;
}

int ihm2mpc_build_tracks(ihm2mpc_handle *h, int32_t max_seg, const int32_t *nseg, const double *coeffs_X, const double *coeffs_Y)
{
    CHECK_H(h);
    if (!nseg || !coeffs_X || !coeffs_Y) return fail("null argument");
    if (h->cfg.nknots % 3 != 0) return fail("nknots = %d is not three laps of samples", h->cfg.nknots);
    if (max_seg < 2) return fail("max_seg must be >= 2");
    for (int t = 0; t < h->cfg.ntracks; t++)
        if (nseg[t] < 2 || nseg[t] > max_seg) return fail("track %d has %d spline segments (2 .. max_seg = %d)", t, nseg[t], max_seg);
    const size_t nc = (size_t)h->cfg.ntracks * max_seg;
    double *dev = nullptr;
    int32_t *dseg = nullptr;
    HIP_TRY(hipMalloc((void **)&dev, nc * 9 * sizeof(double)));
    if (hipMalloc((void **)&dseg, h->cfg.ntracks * sizeof(int32_t)) != hipSuccess) { (void)hipFree(dev); return fail("out of device memory"); }
    double *cX = dev, *cY = dev + nc * 4, *work = dev + nc * 8;
    int rc = 0;
    if (hipMemcpyAsync(cX, coeffs_X, nc * 4 * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(cY, coeffs_Y, nc * 4 * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(dseg, nseg, h->cfg.ntracks * sizeof(int32_t), hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = fail("upload of the spline coefficients failed");
    if (!rc) {
        ihm2_launch_build_tracks(h, max_seg, dseg, cX, cY, work);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) rc = fail("track-table kernels failed");
    }
    (void)hipFree(dev); (void)hipFree(dseg);
    if (rc) return rc;
    h->tracks_set = true; h->geometry_set = true;
    return 0;
}

int ihm2mpc_fit_tracks(ihm2mpc_handle *h, int32_t max_pts, const int32_t *npts, const double *xy, double curv_weight, double *coeffs_X, double *coeffs_Y)
{
    CHECK_H(h);
    if (!npts || !xy || !coeffs_X || !coeffs_Y) return fail("null argument");
    if (!(curv_weight >= 0.0)) return fail("curv_weight must be >= 0");
    if (max_pts < 3 || 7 * max_pts + 2 > 1280) return fail("max_pts must be in 3 .. 182 (the elimination lists the non-zeros of a row in LDS)");
    const int nt = h->cfg.ntracks;
    for (int t = 0; t < nt; t++)
        if (npts[t] < 3 || npts[t] > max_pts) return fail("track %d has %d centre-line points (3 .. max_pts = %d)", t, npts[t], max_pts);
    const size_t m = 7 * (size_t)max_pts, nwork = (size_t)nt * m * (m + 2), nxy = (size_t)nt * max_pts * 2, nc = (size_t)nt * max_pts * 4;
    double *dev = nullptr;
    int32_t *di = nullptr;
    HIP_TRY(hipMalloc((void **)&dev, (nwork + nxy + 2 * nc) * sizeof(double)));
    if (hipMalloc((void **)&di, 2 * (size_t)nt * sizeof(int32_t)) != hipSuccess) { (void)hipFree(dev); return fail("out of device memory"); }
    double *work = dev, *dxy = dev + nwork, *cX = dxy + nxy, *cY = cX + nc;
    std::vector<int32_t> flags(nt, 1);
    int rc = 0;
    if (hipMemcpyAsync(dxy, xy, nxy * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(di, npts, nt * sizeof(int32_t), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemsetAsync(cX, 0, 2 * nc * sizeof(double), h->stream) != hipSuccess) rc = fail("upload of the centre lines failed");
    if (!rc) {
        ihm2_launch_track_fit(h, max_pts, di, dxy, curv_weight, work, cX, cY, di + nt);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(coeffs_X, cX, nc * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipMemcpyAsync(coeffs_Y, cY, nc * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipMemcpyAsync(flags.data(), di + nt, nt * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) rc = fail("spline-fit kernel failed");
    }
    (void)hipFree(dev); (void)hipFree(di);
    if (rc) return rc;
    for (int t = 0; t < nt; t++)
        if (flags[t]) return fail("the spline fit of track %d is singular (coincident centre-line points?)", t);
    return 0;
}

int ihm2mpc_get_tracks(ihm2mpc_handle *h, double *s_ref, double *kappa_ref, double *X_ref, double *Y_ref, double *phi_ref)
{
    CHECK_H(h);
    if (!h->tracks_set) return fail("no track tables yet");
    const size_t n = (size_t)h->cfg.ntracks * h->cfg.nknots * sizeof(double);
    const double *src[5] = {h->s_ref, h->kappa_ref, h->X_ref, h->Y_ref, h->phi_ref};
    double *dst[5] = {s_ref, kappa_ref, X_ref, Y_ref, phi_ref};
    for (int i = 0; i < 5; i++)
        if (dst[i]) {
            if (i >= 2 && !h->geometry_set) return fail("ihm2mpc_set_track_geometry / ihm2mpc_build_tracks has not been called");
            HIP_TRY(hipMemcpyAsync(dst[i], src[i], n, hipMemcpyDeviceToHost, h->stream));
        }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int ihm2mpc_set_track_id(ihm2mpc_handle *h, const int32_t *track_id)
{
    CHECK_H(h);
    if (!track_id) return fail("null argument");
    for (int b = 0; b < h->B; b++)
        if (track_id[b] < 0 || track_id[b] >= h->cfg.ntracks) return fail("track_id[%d] = %d out of range", b, track_id[b]);
    HIP_TRY(hipMemcpyAsync(h->track_id, track_id, (size_t)h->B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int ihm2mpc_set_weights(ihm2mpc_handle *h, const double *W, const double *W_e)
{
    CHECK_H(h);
    if (!W || !W_e) return fail("null argument");
    const int N = h->N, NS = h->NS;
    double V[NY][NZ];
    cost_selector(V);
    std::vector<double> Hs((size_t)NS * 100, 0.0), Gy((size_t)NS * 120, 0.0);
    const double cs = h->cfg.cost_scale_stage;
    for (int k = 0; k < N; k++) {
        const double *Wk = W + (size_t)k * NY * NY;
        double VtW[NZ][NY];
        for (int i = 0; i < NZ; i++)
            for (int j = 0; j < NY; j++) {
                double acc = 0;
                for (int l = 0; l < NY; l++) acc += V[l][i] * Wk[l * NY + j];
                VtW[i][j] = acc;
                Gy[((size_t)k * 10 + i) * 12 + j] = cs * acc;
            }
        for (int i = 0; i < NZ; i++)
            for (int j = 0; j < NZ; j++) {
                double acc = 0;
                for (int l = 0; l < NY; l++) acc += VtW[i][l] * V[l][j];
                Hs[((size_t)k * 10 + i) * 10 + j] = cs * acc;
            }
    }
    for (int i = 0; i < NX; i++)
        for (int j = 0; j < NX; j++) {
            Hs[((size_t)N * 10 + i) * 10 + j] = W_e[i * NX + j];
            Gy[((size_t)N * 10 + i) * 12 + j] = W_e[i * NX + j];
        }
    Hs[((size_t)N * 10 + 8) * 10 + 8] = 1.0;
    Hs[((size_t)N * 10 + 9) * 10 + 9] = 1.0;
    for (int k = 0; k < NS; k++)
        for (int i = 0; i < NZ; i++)
            for (int j = 0; j < i; j++)
                if (fabs(Hs[((size_t)k * 10 + i) * 10 + j] - Hs[((size_t)k * 10 + j) * 10 + i]) > 1e-12 * (1 + fabs(Hs[((size_t)k * 10 + i) * 10 + j])))
                    return fail("weight matrix of stage %d is not symmetric", k);
    h->uniform_H = true;
    for (int k = 1; k < N && h->uniform_H; k++)
        for (int i = 0; i < 100; i++)
            if (Hs[(size_t)k * 100 + i] != Hs[i]) { h->uniform_H = false; break; }
    for (int k = 1; k < N && h->uniform_H; k++)       // "uniform" covers the gradient map as well: the QP kernel keeps one copy of both
        for (int i = 0; i < 120; i++)
            if (Gy[(size_t)k * 120 + i] != Gy[i]) { h->uniform_H = false; break; }
    if (upload_shared(h, Hs.data(), h->Hs, Hs.size()) || upload_shared(h, Gy.data(), h->Gy, Gy.size())) return -1;
    if (upload_shared(h, W, h->Wd, (size_t)N * 144) || upload_shared(h, W_e, h->Wd + (size_t)N * 144, 64)) return -1;
    h->weights_set = true;
    return 0;
}

// Constraint-slot table of the QP kernel.  A slot is a (stage, row) pair with its finite sides; a row with a SOFT
// side is split into one-sided slots (the slack belongs to one side).  Lane `l` of the instance's wavefront owns the
// entries l, l+64, ...; both halves of a split row go to the same lane (they accumulate into the same LDS words
// without atomics).  Tables with soft sides: the first S entries of a lane hold ONE-SIDED slots only -- its soft ones first, then hard
// one-sided ones, else padding -- where S is the NSOFT of the kernel instantiation that will run the table: the kernel keeps slack
// registers, and one side's registers only, for those (kernels_qp.hip: ONE_SIDED).  All-hard tables: every entry may be two-sided (S = 0).
static int rebuild_slots(ihm2mpc_handle *h)
{
    const int NS = h->NS;
    struct Slot { int kc; double lb, ub, zw, Zw; };
    struct Row { int kc; double lb, ub, szl, sZl, szu, sZu; bool fl, fu, sl, su; };
    // the rows with a finite side, stage-major (row 14: the lateral-acceleration row of the stages 1..N-1, kept beside the NC = 14 rows of the tables)
    std::vector<Row> rows;
    int m_act = 0, soft_total = 0;
    for (int k = 0; k < NS; k++)
        for (int c = 0; c < NC + 1; c++) {
            const bool extra = c == NC;
            if (extra && !(h->alat_on && k >= 1 && k < NS - 1)) continue;
            Row r;
            r.kc = k * 16 + c;
            r.lb = extra ? h->alat_lb : h->host_lb[k * NC + c]; r.ub = extra ? h->alat_ub : h->host_ub[k * NC + c];
            r.fl = std::isfinite(r.lb); r.fu = std::isfinite(r.ub);
            if (!r.fl && !r.fu) continue;
            r.szl = extra ? h->alat_sz[0] : h->host_sz[k * NLAM + c]; r.sZl = extra ? h->alat_sZ[0] : h->host_sZ[k * NLAM + c];
            r.szu = extra ? h->alat_sz[1] : h->host_sz[k * NLAM + NC + c]; r.sZu = extra ? h->alat_sZ[1] : h->host_sZ[k * NLAM + NC + c];
            r.sl = r.fl && r.sZl >= 0.0; r.su = r.fu && r.sZu >= 0.0;
            rows.push_back(r);
            soft_total += (int)r.sl + (int)r.su;
            m_act += (int)r.fl + (int)r.fu + (int)r.sl + (int)r.su;
        }
    // (NSOFT, NSLOT) of the kernel instantiations that take this kind of table, in the launchers' order (kernels_qp.hip)
    struct Cand { int S, NSL; };
    std::vector<Cand> cands;
    if (soft_total == 0) cands = {{0, 8}};
    else if (h->alat_on) cands = {{4, 10}};
    else if (h->path_on) cands = {{3, 8}, {4, 10}};
    else cands = {{2, 8}, {4, 10}};
    std::vector<Slot> lanes[64];
    int per_lane = 0, soft_lane = 0, total = 0, S_used = 0;
    bool placed = false;
    for (const Cand &cd : cands) {
        const int S = cd.S;
        std::vector<Slot> ones[64], twos[64];       // one-sided slots (soft ones first), two-sided slots
        int nsoft[64] = {0};
        auto tail = [&](int l) { return (int)twos[l].size() + std::max(0, (int)ones[l].size() - S); };     // entries behind the first S
        // two passes: rows with a soft side first, spread by soft count, then the hard rows by the entries behind the leading ones
        // (all-hard tables, S = 0: by total count -- round-robin)
        for (int pass = 0; pass < 2; pass++)
            for (const Row &r : rows) {
                if ((r.sl || r.su) != (pass == 0)) continue;
                int best = 0;       // least-loaded lane; ties -> lowest lane
                for (int l = 1; l < 64; l++) {
                    const bool fewer_soft = nsoft[l] < nsoft[best], same_soft = nsoft[l] == nsoft[best];
                    const int tl = tail(l) + ((S > 0 && pass == 1 && !(r.fl && r.fu) && (int)ones[l].size() < S) ? -1 : 0);
                    const int tb = tail(best) + ((S > 0 && pass == 1 && !(r.fl && r.fu) && (int)ones[best].size() < S) ? -1 : 0);
                    const bool fewer = (pass == 0) ? ones[l].size() + twos[l].size() < ones[best].size() + twos[best].size() : tl < tb;
                    if (pass == 0 ? (fewer_soft || (same_soft && fewer)) : fewer) best = l;
                }
                if (!r.sl && !r.su) {
                    if (S > 0 && !(r.fl && r.fu)) ones[best].push_back({r.kc, r.lb, r.ub, 0.0, -1.0});     // a hard one-sided row may lead
                    else twos[best].push_back({r.kc, r.lb, r.ub, 0.0, -1.0});
                } else {
                    // the soft half in front of a hard half
                    Slot lo = {r.kc, r.lb, INFINITY, r.sl ? r.szl : 0.0, r.sl ? r.sZl : -1.0}, up = {r.kc, -INFINITY, r.ub, r.su ? r.szu : 0.0, r.su ? r.sZu : -1.0};
                    if (r.fl && r.sl) ones[best].push_back(lo);
                    if (r.fu && r.su) ones[best].push_back(up);
                    if (r.fl && !r.sl) ones[best].push_back(lo);
                    if (r.fu && !r.su) ones[best].push_back(up);
                }
                nsoft[best] += (int)r.sl + (int)r.su;
            }
        int pl = 0, sl_max = 0;
        for (int l = 0; l < 64; l++) {
            std::stable_partition(ones[l].begin(), ones[l].end(), [](const Slot &s) { return s.Zw >= 0.0; });
            pl = std::max(pl, S + tail(l));
            sl_max = std::max(sl_max, nsoft[l]);
        }
        if (soft_total > 0 && (sl_max > S || pl > cd.NSL)) continue;
        // lay the lanes out: S leading one-sided entries (padding where a lane has fewer), then the rest
        total = 0;
        for (int l = 0; l < 64; l++) {
            lanes[l].clear();
            const int lead = std::min(S, (int)ones[l].size());
            for (int i = 0; i < lead; i++) lanes[l].push_back(ones[l][i]);
            if (!twos[l].empty() || (int)ones[l].size() > S)
                for (int i = lead; i < S; i++) lanes[l].push_back({-1, -INFINITY, INFINITY, 0.0, -1.0});
            for (size_t i = S; i < ones[l].size(); i++) lanes[l].push_back(ones[l][i]);
            for (const Slot &t : twos[l]) lanes[l].push_back(t);
            for (const Slot &t : lanes[l]) total += t.kc >= 0;
        }
        per_lane = 0;
        for (int l = 0; l < 64; l++) per_lane = std::max(per_lane, (int)lanes[l].size());
        soft_lane = sl_max; S_used = S; placed = true;
        break;
    }
    h->slots_fit = placed && per_lane * 64 <= MAX_SLOTS && soft_lane <= 4;
    if (!h->slots_fit) return 0;        // ready() reports it: the setters come one by one and a later one may make the rows fit
    if (soft_total > 0) soft_lane = S_used;      // what the launchers select the instantiation by: its NSOFT
    const size_t n = (size_t)per_lane * 64;
    std::vector<int32_t> kc(n, -1);
    std::vector<double> slb(n, -INFINITY), sub(n, INFINITY), zw(n, 0.0), Zw(n, -1.0);
    for (int l = 0; l < 64; l++)
        for (size_t r = 0; r < lanes[l].size(); r++) {
            const Slot &s = lanes[l][r];
            const size_t e = l + 64 * r;
            kc[e] = s.kc; slb[e] = s.lb; sub[e] = s.ub; zw[e] = s.zw; Zw[e] = s.Zw;
        }
    h->nslots = total; h->m_act = m_act; h->nslot_lane = per_lane; h->nsoft_lane = soft_lane;
    // the same rows over 256 lanes for the four-wave latency kernel (all-hard tables: a row is one two-sided slot)
    h->nslot_lane_blk = 0;
    if (soft_lane == 0 && total > 0 && total <= 1024) {
        // entry e of this table IS entry e of the 64-lane table (thread t of the block holds the entries t, t + 256): the slot sums of the
        // four-wave body are taken in the order of the 64-lane table, which makes its results those of the one-wave body bit for bit
        const int per_blk = ((int)n + 255) / 256;
        std::vector<int32_t> kcb((size_t)per_blk * 256, -1);
        std::vector<double> lbb((size_t)per_blk * 256, -INFINITY), ubb((size_t)per_blk * 256, INFINITY);
        for (size_t e = 0; e < n; e++) { kcb[e] = kc[e]; lbb[e] = slb[e]; ubb[e] = sub[e]; }
        HIP_TRY(hipMemcpyAsync(h->slot_kc_blk, kcb.data(), kcb.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (upload_shared(h, lbb.data(), h->slot_lb_blk, lbb.size()) || upload_shared(h, ubb.data(), h->slot_ub_blk, ubb.size())) return -1;
        h->nslot_lane_blk = per_blk;
    }
    if (upload_shared(h, h->host_lb, h->st_lb, (size_t)NS * NC) || upload_shared(h, h->host_ub, h->st_ub, (size_t)NS * NC) ||
        upload_shared(h, h->host_sz, h->st_sz, (size_t)NS * NLAM) || upload_shared(h, h->host_sZ, h->st_sZ, (size_t)NS * NLAM))
        return -1;
    if (n) {
        HIP_TRY(hipMemcpyAsync(h->slot_kc, kc.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (upload_shared(h, slb.data(), h->slot_lb, n) || upload_shared(h, sub.data(), h->slot_ub, n) ||
            upload_shared(h, zw.data(), h->slot_zw, n) || upload_shared(h, Zw.data(), h->slot_Zw, n))
            return -1;
    }
    return 0;
}

int ihm2mpc_set_bounds(ihm2mpc_handle *h, const double *lbx, const double *ubx, const double *lbu, const double *ubu,
                       const double *C, const double *D, const double *lg, const double *ug)
{
    CHECK_H(h);
    if (!lbx || !ubx || !lbu || !ubu || !C || !D || !lg || !ug) return fail("null argument");
    const int N = h->N, NS = h->NS;
    for (int i = 0; i < NS * NX; i++) if (lbx[i] > ubx[i]) return fail("lbx > ubx at flat index %d", i);
    for (int i = 0; i < N * NU; i++) if (lbu[i] > ubu[i]) return fail("lbu > ubu at flat index %d", i);
    for (int i = 0; i < N * NG; i++) if (lg[i] > ug[i]) return fail("lg > ug at flat index %d", i);
    std::vector<double> CD((size_t)N * 20);
    for (int k = 0; k < N; k++)
        for (int r = 0; r < NG; r++) {
            for (int j = 0; j < NX; j++) CD[((size_t)k * 2 + r) * 10 + j] = C[((size_t)k * 2 + r) * NX + j];
            for (int j = 0; j < NU; j++) CD[((size_t)k * 2 + r) * 10 + 8 + j] = D[((size_t)k * 2 + r) * NU + j];
        }
    h->uniform_CD = true;
    for (int k = 1; k < N && h->uniform_CD; k++)
        for (int i = 0; i < 20; i++)
            if (CD[(size_t)k * 20 + i] != CD[i]) { h->uniform_CD = false; break; }
    for (int k = 0; k < NS; k++)
        for (int c = 0; c < 12; c++) {       // rows 12, 13 belong to set_path_constraints
            double lb = -INFINITY, ub = INFINITY;
            if (c < 8) { if (k >= 1) { lb = lbx[k * 8 + c]; ub = ubx[k * 8 + c]; } }
            else if (c < 10) { if (k < N) { lb = lbu[k * 2 + c - 8]; ub = ubu[k * 2 + c - 8]; } }
            else { if (k < N) { lb = lg[k * 2 + c - 10]; ub = ug[k * 2 + c - 10]; } }
            h->host_lb[k * NC + c] = (std::fabs(lb) < 1e20) ? lb : -INFINITY;
            h->host_ub[k * NC + c] = (std::fabs(ub) < 1e20) ? ub : INFINITY;
        }
    if (rebuild_slots(h)) return -1;
    if (upload_shared(h, lbx, h->lbx, (size_t)NS * 8) || upload_shared(h, ubx, h->ubx, (size_t)NS * 8) ||
        upload_shared(h, lbu, h->lbu, (size_t)N * 2) || upload_shared(h, ubu, h->ubu, (size_t)N * 2) ||
        upload_shared(h, CD.data(), h->CD, CD.size()) || upload_shared(h, lg, h->lg, (size_t)N * 2) ||
        upload_shared(h, ug, h->ug, (size_t)N * 2))
        return -1;
    h->bounds_set = true;
    return 0;
}

int ihm2mpc_set_soft(ihm2mpc_handle *h, const double *soft_z, const double *soft_Z)
{
    CHECK_H(h);
    const int n = h->NS * NLAM;
    if ((soft_z == nullptr) != (soft_Z == nullptr)) return fail("soft_z and soft_Z must both be given or both be NULL");
    for (int i = 0; i < n; i++) {
        const double Z = soft_Z ? soft_Z[i] : -1.0, z = soft_z ? soft_z[i] : 0.0;
        if (Z >= 0.0 && !(z >= 0.0)) return fail("soft_z < 0 at flat index %d (the slack penalty must be non-decreasing)", i);
        if (Z >= 0.0 && !(Z + z > 0.0)) return fail("soft side %d has neither a linear nor a quadratic penalty", i);
        h->host_sz[i] = z; h->host_sZ[i] = Z;
    }
    HIP_TRY(hipMemsetAsync(h->slk, 0, (size_t)h->B * h->NS * NLAM * sizeof(double), h->stream));
    if (h->bounds_set && rebuild_slots(h)) return -1;
    return 0;
}

int ihm2mpc_set_path_constraints(ihm2mpc_handle *h, int32_t enable, double car_length, double car_width, const double *widths,
                                 const double *lh, const double *uh)
{
    CHECK_H(h);
    const int NS = h->NS;
    if (enable) {
        if (!widths || !lh || !uh) return fail("null argument");
        if (!(car_length >= 0.0) || !(car_width >= 0.0)) return fail("car_length and car_width must be non-negative");
        for (int i = 0; i < NH; i++) if (lh[i] > uh[i]) return fail("lh > uh at row %d", i);
        for (int t = 0; t < h->cfg.ntracks * 2; t++) if (!(widths[t] > 0.0)) return fail("track width %d is not positive", t);
        if (upload_shared(h, widths, h->widths, (size_t)h->cfg.ntracks * 2)) return -1;
        h->car_L = car_length; h->car_W = car_width;
    }
    h->path_on = enable ? 1 : 0;
    for (int k = 0; k < NS; k++)
        for (int i = 0; i < NH; i++) {
            const bool on = enable && k >= 1;       // x_0 is fixed: the rows of stage 0 are constants
            h->host_lb[k * NC + 12 + i] = (on && std::fabs(lh[i]) < 1e20) ? lh[i] : -INFINITY;
            h->host_ub[k * NC + 12 + i] = (on && std::fabs(uh[i]) < 1e20) ? uh[i] : INFINITY;
        }
    if (h->bounds_set && rebuild_slots(h)) return -1;
    return 0;
}

int ihm2mpc_set_alat_constraint(ihm2mpc_handle *h, int32_t enable, double a_lat_min, double a_lat_max, const double *soft_z, const double *soft_Z)
{
    CHECK_H(h);
    if (enable) {
        if (a_lat_min != a_lat_min || a_lat_max != a_lat_max) return fail("a_lat bound is not a number");
        if (a_lat_min > a_lat_max) return fail("a_lat_min > a_lat_max");
        for (int i = 0; i < 2; i++) {
            const double z = soft_z ? soft_z[i] : 0.0, Z = soft_Z ? soft_Z[i] : -1.0;
            if (z != z || Z != Z) return fail("soft penalty of the a_lat row is not a number");
            if (Z >= 0.0 && !(z >= 0.0)) return fail("soft_z < 0 on the a_lat row (the slack penalty must be non-decreasing)");
            if (Z >= 0.0 && !(Z + z > 0.0)) return fail("a soft side of the a_lat row has neither a linear nor a quadratic penalty");
            h->alat_sz[i] = z; h->alat_sZ[i] = Z;
        }
        h->alat_lb = (std::fabs(a_lat_min) < 1e20) ? a_lat_min : -INFINITY;
        h->alat_ub = (std::fabs(a_lat_max) < 1e20) ? a_lat_max : INFINITY;
    }
    h->alat_on = enable ? 1 : 0;
    HIP_TRY(hipMemsetAsync(h->lam_a, 0, (size_t)h->B * h->NS * 2 * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->slk_a, 0, (size_t)h->B * h->NS * 2 * sizeof(double), h->stream));
    if (h->bounds_set && rebuild_slots(h)) return -1;
    return 0;
}

int ihm2mpc_set_alat_multipliers(ihm2mpc_handle *h, const double *lam, const double *slk)
{
    CHECK_H(h);
    if (lam) { if (upload(h, lam, h->lam_a, h->NS * 2)) return -1; }
    else HIP_TRY(hipMemsetAsync(h->lam_a, 0, (size_t)h->B * h->NS * 2 * sizeof(double), h->stream));
    if (slk) { if (upload(h, slk, h->slk_a, h->NS * 2)) return -1; }
    else HIP_TRY(hipMemsetAsync(h->slk_a, 0, (size_t)h->B * h->NS * 2 * sizeof(double), h->stream));
    return 0;
}

int ihm2mpc_get_alat_multipliers(ihm2mpc_handle *h, double *lam, double *slk)
{
    CHECK_H(h);
    if (lam && download(h, h->lam_a, lam, h->NS * 2)) return -1;
    if (slk && download(h, h->slk_a, slk, h->NS * 2)) return -1;
    return 0;
}

#define SETTER(name, field, elems)                                   \
    int ihm2mpc_set_##name(ihm2mpc_handle *h, const double *v)       \
    {                                                                \
        CHECK_H(h);                                                  \
        if (!v) return fail("null argument");                        \
        return upload(h, v, h->field, (elems));                      \
    }
SETTER(x0, x0, NX)
SETTER(x, x, h->NS * NX)
SETTER(u, u, h->N * NU)
SETTER(yref, yref, h->N * NY)
SETTER(yref_e, yref_e, NX)
#undef SETTER

int ihm2mpc_set_multipliers(ihm2mpc_handle *h, const double *pi, const double *lam)
{
    CHECK_H(h);
    if (pi) { if (upload(h, pi, h->pi, h->NS * NX)) return -1; }
    else HIP_TRY(hipMemsetAsync(h->pi, 0, (size_t)h->B * h->NS * NX * sizeof(double), h->stream));
    if (lam) { if (upload(h, lam, h->lam, h->NS * NLAM)) return -1; }
    else {
        HIP_TRY(hipMemsetAsync(h->lam, 0, (size_t)h->B * h->NS * NLAM * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(h->lam_a, 0, (size_t)h->B * h->NS * 2 * sizeof(double), h->stream));      // "no multipliers" covers row 14 as well
    }
    return 0;
}

int ihm2mpc_set_slacks(ihm2mpc_handle *h, const double *sl)
{
    CHECK_H(h);
    if (sl) return upload(h, sl, h->slk, h->NS * NLAM);
    HIP_TRY(hipMemsetAsync(h->slk, 0, (size_t)h->B * h->NS * NLAM * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->slk_a, 0, (size_t)h->B * h->NS * 2 * sizeof(double), h->stream));
    return 0;
}

int ihm2mpc_set_stage(ihm2mpc_handle *h, int32_t instance, int32_t stage, const char *field, const double *value, int32_t n)
{
    CHECK_H(h);
    FieldInfo fi;
    if (field_info(h, field, &fi)) return -1;
    if (!value) return fail("null argument");
    if (instance < 0 || instance >= h->B) return fail("instance %d out of range", instance);
    const std::string f(field);
    if ((f == "lbx" || f == "ubx") && stage != 0)
        return fail("per-instance '%s' exists at stage 0 only (the initial state); stage bounds are shared: ihm2mpc_set_bounds", field);
    if (f == "yref_e") stage = 0;
    if (stage < 0 || stage >= fi.nstages) return fail("stage %d out of range for field '%s'", stage, field);
    if (n != fi.per_stage) return fail("field '%s' has %d entries per stage, got %d", field, fi.per_stage, n);
    double *dst = fi.base + ((size_t)instance * fi.nstages + stage) * fi.per_stage;
    HIP_TRY(hipMemcpyAsync(dst, value, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int ihm2mpc_get_stage(ihm2mpc_handle *h, int32_t instance, int32_t stage, const char *field, double *value, int32_t n)
{
    CHECK_H(h);
    FieldInfo fi;
    if (field_info(h, field, &fi)) return -1;
    if (!value) return fail("null argument");
    if (instance < 0 || instance >= h->B) return fail("instance %d out of range", instance);
    if (std::string(field) == "yref_e") stage = 0;
    if (stage < 0 || stage >= fi.nstages) return fail("stage %d out of range for field '%s'", stage, field);
    if (n != fi.per_stage) return fail("field '%s' has %d entries per stage, got %d", field, fi.per_stage, n);
    const double *src = fi.base + ((size_t)instance * fi.nstages + stage) * fi.per_stage;
    HIP_TRY(hipMemcpyAsync(value, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int ihm2mpc_init_guess(ihm2mpc_handle *h, double v_ref_scale)
{
    CHECK_H(h);
    if (ready(h)) return -1;
    ihm2_launch_init_guess(h, v_ref_scale, 0);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ihm2mpc_reinit_failed(ihm2mpc_handle *h, double v_ref_scale)
{
    CHECK_H(h);
    if (ready(h)) return -1;
    ihm2_launch_init_guess(h, v_ref_scale, 1);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ihm2mpc_prepare_step(ihm2mpc_handle *h, double s_target)
{
    CHECK_H(h);
    if (h->lap_wrap) ihm2_launch_wrap_lap(h);
    ihm2_launch_prepare(h, s_target, 3, h->stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ihm2mpc_linearize(ihm2mpc_handle *h)
{
    CHECK_H(h);
    if (ready(h)) return -1;
    ihm2_launch_linearize(h);
    HIP_TRY(hipGetLastError());
    return 0;
}

// SQP mode: buffers of the line search, allocated on first use (36 KB per instance)
static int sqp_buffers(ihm2mpc_handle *h)
{
    if (h->ls_x) return 0;
    const size_t B = h->B, N = h->N, NS = h->NS;
#define DA(p, n) if (dalloc(&h->p, (n))) return -1
    DA(ls_x, B * NS * 8); DA(ls_u, B * N * 2); DA(ls_pi, B * NS * 8); DA(ls_lam, B * NS * NLAM); DA(ls_slk, B * NS * NLAM);
    DA(ls_wpi, B * NS * 8); DA(ls_wlam, B * NS * NLAM); DA(ls_alpha, B); DA(ls_args, 64); DA(ls_done, B); DA(ls_status, B); DA(ls_iter, B); DA(ls_qp_acc, B);
#undef DA
    return 0;
}

// SQP mode with the collocation integrator: room for the line search's trial-point rollouts (every step length of the ladder)
// Trial step lengths of the backtracking ladder 1, rho, rho^2, ... >= alpha_min -- the same floating-point sequence as the loop of
// line_search_body (sqp_body.hpp).  ihm2mpc_set_sqp_options refuses option pairs whose ladder is longer than IHM2MPC_LS_MAX_TRIALS:
// one length everywhere (the buffer of the collocation rollouts, the rollout launches, the line search).
#define IHM2MPC_LS_MAX_TRIALS 64
static int ladder_length(double alpha_min, double alpha_red)
{
    int n = 1;
    for (double al = alpha_red; al >= alpha_min && n <= IHM2MPC_LS_MAX_TRIALS; al *= alpha_red) n++;
    return n;
}

static int sqp_phi_buffer(ihm2mpc_handle *h, int *n_alpha_out)
{
    const size_t B = h->B, N = h->N;
    const int n_alpha = ladder_length(h->sqp_alpha_min, h->sqp_alpha_red);
    if (n_alpha > IHM2MPC_LS_MAX_TRIALS) return fail("backtracking ladder longer than %d trial steps", IHM2MPC_LS_MAX_TRIALS);
    if (!h->ls_phi || h->ls_nalpha < n_alpha) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->ls_phi) (void)hipFree(h->ls_phi);
        h->ls_phi = nullptr;
        HIP_TRY(hipMalloc((void **)&h->ls_phi, (size_t)n_alpha * B * N * 8 * sizeof(double)));
        h->ls_nalpha = n_alpha;
    }
    if (!h->ls_pending) HIP_TRY(hipMalloc((void **)&h->ls_pending, (size_t)B * sizeof(int32_t)));
    if (n_alpha_out) *n_alpha_out = n_alpha;
    return 0;
}

// SQP mode: n_iter iterations of [copy the iterate aside -> linearise -> QP -> convergence test + line search].  join: the first
// QP waits for ev_join (ihm2mpc_step runs the plant and the reference ramp beside the first linearisation).
static int sqp_iterations(ihm2mpc_handle *h, int n_iter, bool join)
{
    if (sqp_buffers(h)) return -1;
    const size_t B = h->B, N = h->N;
    HIP_TRY(hipMemsetAsync(h->ls_done, 0, B * sizeof(int32_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->ls_iter, 0, B * sizeof(int32_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->ls_qp_acc, 0, B * sizeof(int32_t), h->stream));
    for (int it = 0; it < n_iter; it++) {
        // the iterate the QP is built at: the line search walks from it towards the QP's full step
        ihm2_launch_copy_iterate(h);
        ihm2_launch_linearize(h);
        if (it == 0 && join) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
        if (it == n_iter - 1) HIP_TRY(hipEventRecord(h->ev[1], h->stream));
        if (ihm2_launch_qp(h)) return fail("problem exceeds the QP kernel limits (LDS or constraint slots)");
        if (h->cfg.integrator_type != IHM2MPC_INTEG_ERK && h->sqp_globalization) {
            // collocation integrator: the rollouts of the line search's trial points (all step lengths of the ladder) in their own launch
            int n_alpha = 1;
            if (sqp_phi_buffer(h, &n_alpha)) return -1;
            // most instances accept one of the first step lengths: those rollouts for everybody, the rest of the ladder only for the
            // instances the first line-search launch leaves open (which then redo their ladder: same arithmetic as one launch)
            // (a few instances -- the whole ladder is one round of wavefronts on the chip -- keep the single launch: two launches less per iteration)
            const int j_first = (n_alpha < 3 || (long)n_alpha * (long)B * (long)N <= 16384) ? n_alpha : 3;
            ihm2_launch_rollout_irk(h, 0, j_first, h->ls_phi, nullptr);
            if (j_first < n_alpha) {
                ihm2_launch_line_search(h, it, it == n_iter - 1, 1, j_first);
                ihm2_launch_rollout_irk(h, j_first, n_alpha, h->ls_phi, h->ls_pending);
                ihm2_launch_line_search(h, it, it == n_iter - 1, 2, 0);
                continue;
            }
        }
        ihm2_launch_line_search(h, it, it == n_iter - 1);
    }
    return 0;
}

static int sqp_iter_count(ihm2mpc_handle *h) { return h->cfg.nlp_solver_max_iter > 0 ? h->cfg.nlp_solver_max_iter : 1; }

int ihm2mpc_solve(ihm2mpc_handle *h, int32_t n_iter)
{
    CHECK_H(h);
    if (ready(h)) return -1;
    const bool sqp = h->cfg.nlp_solver_type == IHM2MPC_SQP;
    if (n_iter <= 0) n_iter = sqp ? sqp_iter_count(h) : 1;
    HIP_TRY(hipEventRecord(h->ev[0], h->stream));
    if (sqp) {
        if (sqp_iterations(h, n_iter, false)) return -1;
    } else {
        for (int it = 0; it < n_iter; it++) {
            ihm2_launch_linearize(h);
            if (it == n_iter - 1) HIP_TRY(hipEventRecord(h->ev[1], h->stream));
            if (ihm2_launch_qp(h)) return fail("problem exceeds the QP kernel limits (LDS or constraint slots)");
        }
    }
    HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

int ihm2mpc_set_sqp_options(ihm2mpc_handle *h, int32_t globalization, double alpha_min, double alpha_reduction, double eps_sufficient_descent,
                            int32_t use_sufficient_descent, int32_t full_step_dual, const double *tol)
{
    CHECK_H(h);
    if (globalization != IHM2MPC_FIXED_STEP && globalization != IHM2MPC_MERIT_BACKTRACKING) return fail("unknown globalization %d", globalization);
    if (!(alpha_min > 0.0 && alpha_min <= 1.0)) return fail("alpha_min must be in (0, 1]");
    if (!(alpha_reduction > 0.0 && alpha_reduction < 1.0)) return fail("alpha_reduction must be in (0, 1)");
    if (!(eps_sufficient_descent >= 0.0 && eps_sufficient_descent < 1.0)) return fail("eps_sufficient_descent must be in [0, 1)");
    if (globalization == IHM2MPC_MERIT_BACKTRACKING && ladder_length(alpha_min, alpha_reduction) > IHM2MPC_LS_MAX_TRIALS)
        return fail("alpha_reduction %g with alpha_min %g makes a backtracking ladder of more than %d trial steps", alpha_reduction, alpha_min, IHM2MPC_LS_MAX_TRIALS);
    h->sqp_globalization = globalization; h->sqp_alpha_min = alpha_min; h->sqp_alpha_red = alpha_reduction; h->sqp_eps = eps_sufficient_descent;
    h->sqp_use_suff = use_sufficient_descent ? 1 : 0; h->sqp_full_step_dual = full_step_dual ? 1 : 0;
    if (tol)
        for (int i = 0; i < 4; i++) {
            if (!(tol[i] >= 0.0)) return fail("tolerance %d is negative", i);
            h->sqp_tol[i] = tol[i];
        }
    return 0;
}

int ihm2mpc_get_sqp_stats(ihm2mpc_handle *h, int32_t *sqp_iter, double *alpha)
{
    CHECK_H(h);
    if (!h->ls_x) return fail("no SQP-mode solve has run on this handle");
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (sqp_iter) HIP_TRY(hipMemcpy(sqp_iter, h->ls_iter, (size_t)h->B * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (alpha) HIP_TRY(hipMemcpy(alpha, h->ls_alpha, (size_t)h->B * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int ihm2mpc_get_timings(ihm2mpc_handle *h, double *ms, int32_t n)
{
    CHECK_H(h);
    if (!ms || n < 3) return fail("need room for 3 values");
    HIP_TRY(hipEventSynchronize(h->ev[2]));
    float t_total = 0, t_qp = 0;
    HIP_TRY(hipEventElapsedTime(&t_total, h->ev[0], h->ev[2]));
    HIP_TRY(hipEventElapsedTime(&t_qp, h->ev[1], h->ev[2]));
    ms[0] = t_total; ms[2] = t_qp; ms[1] = t_total - t_qp;
    return 0;
}

int ihm2mpc_get_linearization(ihm2mpc_handle *h, double *A, double *Bm, double *b)
{
    CHECK_H(h);
    const size_t n = (size_t)h->B * h->N;
    std::vector<double> rec(n * LIN_REC);
    HIP_TRY(hipMemcpyAsync(rec.data(), h->lin, rec.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (size_t r = 0; r < n; r++) {
        if (A) memcpy(A + r * 64, rec.data() + r * LIN_REC, 64 * sizeof(double));
        if (Bm) memcpy(Bm + r * 16, rec.data() + r * LIN_REC + 64, 16 * sizeof(double));
        if (b) memcpy(b + r * 8, rec.data() + r * LIN_REC + 80, 8 * sizeof(double));
    }
    return 0;
}

#define GETTER(name, field, elems)                             \
    int ihm2mpc_get_##name(ihm2mpc_handle *h, double *v)       \
    {                                                          \
        CHECK_H(h);                                            \
        if (!v) return fail("null argument");                  \
        return download(h, h->field, v, (elems));              \
    }
GETTER(x, x, h->NS * NX)
GETTER(u, u, h->N * NU)
GETTER(u0, u0, NU)
GETTER(residuals, res, 4)
GETTER(qp_residuals, qp_res, 4)
GETTER(x0, x0, NX)
GETTER(slacks, slk, h->NS * NLAM)
#undef GETTER

int ihm2mpc_get_multipliers(ihm2mpc_handle *h, double *pi, double *lam)
{
    CHECK_H(h);
    if (pi && download(h, h->pi, pi, h->NS * NX)) return -1;
    if (lam && download(h, h->lam, lam, h->NS * NLAM)) return -1;
    return 0;
}

int ihm2mpc_get_status(ihm2mpc_handle *h, int32_t *status)
{
    CHECK_H(h);
    if (!status) return fail("null argument");
    HIP_TRY(hipMemcpyAsync(status, h->status, (size_t)h->B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int ihm2mpc_get_qp_iter(ihm2mpc_handle *h, int32_t *qp_iter)
{
    CHECK_H(h);
    if (!qp_iter) return fail("null argument");
    HIP_TRY(hipMemcpyAsync(qp_iter, h->qp_iter, (size_t)h->B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// ---- device-pointer variants: instance-major device buffers, no host round trip ----
#define DEVCOPY(name, dst, src, bytes)                                                                  \
    int ihm2mpc_##name(ihm2mpc_handle *h, DEVCOPY_ARG dptr)                                             \
    {                                                                                                   \
        CHECK_H(h);                                                                                     \
        if (!dptr) return fail("null argument");                                                        \
        HIP_TRY(hipMemcpyAsync((void *)(dst), (const void *)(src), (bytes), hipMemcpyDeviceToDevice, h->stream)); \
        return 0;                                                                                       \
    }
#define DEVCOPY_ARG const void *
DEVCOPY(set_x0_device, h->x0, dptr, (size_t)h->B * NX * sizeof(double))
#undef DEVCOPY_ARG
#define DEVCOPY_ARG void *
DEVCOPY(get_u0_device, dptr, h->u0, (size_t)h->B * NU * sizeof(double))
DEVCOPY(get_x_device, dptr, h->x, (size_t)h->B * h->NS * NX * sizeof(double))
DEVCOPY(get_u_device, dptr, h->u, (size_t)h->B * h->N * NU * sizeof(double))
DEVCOPY(get_status_device, dptr, h->status, (size_t)h->B * sizeof(int32_t))
#undef DEVCOPY_ARG
#undef DEVCOPY

// ---- plant ----
int ihm2mpc_sim_step(ihm2mpc_handle *h, int32_t model, int32_t M_sim, const double *x, const double *u, double *x_next)
{
    CHECK_H(h);
    if (!h->tracks_set) return fail("ihm2mpc_set_tracks has not been called");
    if (!x || !u || !x_next) return fail("null argument");
    if (M_sim < 1) return fail("M_sim must be >= 1");
    if (h->cfg.sim_integrator_type == IHM2MPC_INTEG_ERK && rk4_unstable(h->cfg.dt, M_sim)) return fail("RK4 with M_sim = %d sub-steps of dt = %g is unstable on the actuator lags: use M_sim >= %d", M_sim, h->cfg.dt, (int)ceil(h->cfg.dt / (2.78 * 1e-3)));
    if (model < -2 || model > IHM2MPC_MODEL_FDYN6U) return fail("unknown plant model %d", model);
    double *xs = h->scratch, *us = h->scratch + (size_t)h->B * 8, *xn = h->scratch + (size_t)h->B * 16;
    if (upload(h, x, xs, NX) || upload(h, u, us, NU)) return -1;
    ihm2_launch_sim(h, model, M_sim, xs, us, xn, h->stream, nullptr);
    HIP_TRY(hipGetLastError());
    return download(h, xn, x_next, NX);
}

int ihm2mpc_sim_advance(ihm2mpc_handle *h, int32_t model, int32_t M_sim)
{
    CHECK_H(h);
    if (!h->tracks_set) return fail("ihm2mpc_set_tracks has not been called");
    if (M_sim < 1) return fail("M_sim must be >= 1");
    if (h->cfg.sim_integrator_type == IHM2MPC_INTEG_ERK && rk4_unstable(h->cfg.dt, M_sim)) return fail("RK4 with M_sim = %d sub-steps of dt = %g is unstable on the actuator lags: use M_sim >= %d", M_sim, h->cfg.dt, (int)ceil(h->cfg.dt / (2.78 * 1e-3)));
    if (model < -2 || model > IHM2MPC_MODEL_FDYN6U) return fail("unknown plant model %d", model);
    ihm2_launch_sim(h, model, M_sim, h->x0, h->u0, h->x0, h->stream, h->active_set ? h->active : nullptr);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ihm2mpc_host_alloc(uint64_t nbytes, void **p)
{
    if (!p || nbytes == 0) return fail("null argument");
    HIP_TRY(hipHostMalloc(p, nbytes, hipHostMallocDefault));
    return 0;
}

int ihm2mpc_host_free(void *p)
{
    if (p) HIP_TRY(hipHostFree(p));
    return 0;
}

// IHM2Controller.compute_control (python/main.py:297-334) in one call: x0 in, reference ramp + warm-start shift, one solve
// (RTI or the configured SQP iterations), u0 and status out -- one host-device round trip, one wait.
int ihm2mpc_compute_control(ihm2mpc_handle *h, const double *x0, double s_target, double *u0, int32_t *status)
{
    CHECK_H(h);
    if (!x0 || !u0) return fail("null argument");
    if (ready(h)) return -1;
    const size_t B = h->B;
    HIP_TRY(hipMemcpyAsync(h->x0, x0, B * NX * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->lap_wrap) ihm2_launch_wrap_lap(h);
    ihm2_launch_prepare(h, s_target, 3, h->stream);
    if (ihm2mpc_solve(h, 0)) return -1;
    HIP_TRY(hipMemcpyAsync(u0, h->u0, B * NU * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (status) HIP_TRY(hipMemcpyAsync(status, h->status, B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int ihm2mpc_get_u0_async(ihm2mpc_handle *h, double *pinned_dst)
{
    CHECK_H(h);
    if (!pinned_dst) return fail("null argument");
    HIP_TRY(hipMemcpyAsync(pinned_dst, h->u0, (size_t)h->B * NU * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return 0;
}

int ihm2mpc_set_lap_wrap(ihm2mpc_handle *h, int32_t enable)
{
    CHECK_H(h);
    h->lap_wrap = enable != 0;
    return 0;
}

int ihm2mpc_set_active(ihm2mpc_handle *h, const int32_t *active)
{
    CHECK_H(h);
    h->active_set = active != nullptr;
    h->freeze_armed = false;      // set_active(NULL) also forgets the mask a frozen loop left behind
    if (active) {
        HIP_TRY(hipMemcpyAsync(h->active, active, (size_t)h->B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return 0;
}

int ihm2mpc_step(ihm2mpc_handle *h, int32_t model, int32_t M_sim, double s_target)
{
    CHECK_H(h);
    if (ready(h)) return -1;
    if (M_sim < 1) return fail("M_sim must be >= 1");
    if (h->cfg.sim_integrator_type == IHM2MPC_INTEG_ERK && rk4_unstable(h->cfg.dt, M_sim)) return fail("RK4 with M_sim = %d sub-steps of dt = %g is unstable on the actuator lags: use M_sim >= %d", M_sim, h->cfg.dt, (int)ceil(h->cfg.dt / (2.78 * 1e-3)));
    if (model < -2 || model > IHM2MPC_MODEL_FDYN6U) return fail("unknown plant model %d", model);
    // The plant step and the reference ramp only feed the QP (through x0 and yref); the warm-start shift and the
    // linearisation only need the previous iterate.  Two branches, joined in front of the QP kernel.
    if (h->lap_wrap) ihm2_launch_wrap_lap(h);
    HIP_TRY(hipEventRecord(h->ev[0], h->stream));
    HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
    ihm2_launch_sim(h, model, M_sim, h->x0, h->u0, h->x0, h->stream2, h->active_set ? h->active : nullptr);
    ihm2_launch_prepare(h, s_target, 1, h->stream2);
    HIP_TRY(hipEventRecord(h->ev_join, h->stream2));
    ihm2_launch_prepare(h, s_target, 2, h->stream);
    if (h->cfg.nlp_solver_type == IHM2MPC_SQP) {
        if (sqp_iterations(h, sqp_iter_count(h), true)) return -1;
    } else {
        ihm2_launch_linearize(h);
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
        HIP_TRY(hipEventRecord(h->ev[1], h->stream));
        if (ihm2_launch_qp(h)) return fail("problem exceeds the QP kernel limits (LDS or constraint slots)");
    }
    HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

// room for the histories of ihm2mpc_run_steps: growing them costs a few device allocations, which a caller that times its
// run_steps calls wants to have behind it
int ihm2mpc_reserve_history(ihm2mpc_handle *h, int32_t n_steps)
{
    CHECK_H(h);
    if (n_steps < 1) return fail("n_steps must be >= 1");
    const size_t B = h->B, n = n_steps;
    if (h->hist_cap >= n) return 0;
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (void *p : {(void *)h->hist_u0, (void *)h->hist_x0, (void *)h->hist_st, (void *)h->hist_it}) if (p) (void)hipFree(p);
    h->hist_u0 = h->hist_x0 = nullptr; h->hist_st = h->hist_it = nullptr; h->hist_cap = 0;
    if (dalloc(&h->hist_u0, n * B * 2) || dalloc(&h->hist_x0, n * B * 8) || dalloc(&h->hist_st, n * B) || dalloc(&h->hist_it, n * B)) return -1;
    h->hist_cap = n;
    return 0;
}

// n_steps control steps of the MiL loop with everything on the device (python/main.py:476-517).  Where the configuration has a
// persistent instantiation (fkin6 OCP, RTI, all-hard constraint table) this is ONE launch in which every instance runs its
// steps back to back; otherwise n_steps x ihm2mpc_step.  Same results either way.
int ihm2mpc_run_steps(ihm2mpc_handle *h, int32_t model, int32_t M_sim, double s_target, int32_t n_steps, int32_t freeze, double lap_stop,
                      double *u0_hist, double *x0_hist, int32_t *status_hist, int32_t *qp_iter_hist)
{
    CHECK_H(h);
    if (ready(h)) return -1;
    if (M_sim < 1) return fail("M_sim must be >= 1");
    if (h->cfg.sim_integrator_type == IHM2MPC_INTEG_ERK && rk4_unstable(h->cfg.dt, M_sim)) return fail("RK4 with M_sim = %d sub-steps of dt = %g is unstable on the actuator lags: use M_sim >= %d", M_sim, h->cfg.dt, (int)ceil(h->cfg.dt / (2.78 * 1e-3)));
    if (model < -2 || model > IHM2MPC_MODEL_FDYN6U) return fail("unknown plant model %d", model);
    if (n_steps < 1) return fail("n_steps must be >= 1");
    const size_t B = h->B, n = n_steps;
    if (ihm2mpc_reserve_history(h, n_steps)) return -1;
    if (freeze && !h->active_set && !h->freeze_armed) {       // every car starts driving; later calls keep the mask the device updated
        std::vector<int32_t> ones(B, 1);
        HIP_TRY(hipMemcpyAsync(h->active, ones.data(), B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->freeze_armed = true;
    }
    // The persistent loop pays off while every instance has a wavefront slot of its own (the QP's 40 KB of LDS allow 4 per CU):
    // a larger batch would run in rounds of whole n_steps-long loops, whereas launches per step backfill the slots of finished
    // QPs with the next instances.
    // (IHM2MPC_PERSISTENT_ROUNDS=1: take the one-launch loop for larger batches too -- the workgroups then run in rounds of whole histories;
    // measured round 4 against launches per step for the dynamic models, NOTES.md R4)
    static const bool rounds = [] { const char *e = getenv("IHM2MPC_PERSISTENT_ROUNDS"); return e && e[0] == '1'; }();
    const bool resident = B <= (size_t)4 * h->n_cu || (rounds && !freeze);
    int rc = 1;
    if (h->cfg.nlp_solver_type == IHM2MPC_SQP && sqp_buffers(h)) return -1;
    if (h->cfg.nlp_solver_type == IHM2MPC_SQP && h->cfg.integrator_type != IHM2MPC_INTEG_ERK && h->sqp_globalization && sqp_phi_buffer(h, nullptr)) return -1;
    if (resident) {
        HIP_TRY(hipEventRecord(h->ev[0], h->stream));
        HIP_TRY(hipEventRecord(h->ev[1], h->stream));
        rc = ihm2_launch_steps(h, model, M_sim, s_target, n_steps, freeze ? 1 : 0, lap_stop, h->hist_u0, h->hist_x0, h->hist_st, h->hist_it);
        if (rc == 0) HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    }
    if (rc != 0) {
        if (freeze) return fail("no persistent loop for this configuration (needs batch-shared weights and rows for soft tables or the collocation integrator, and a batch of at most %d): call ihm2mpc_step per control period", 4 * h->n_cu);
        for (size_t i = 0; i < n; i++) {      // launches per step, histories by device-to-device copies in stream order
            if (ihm2mpc_step(h, model, M_sim, s_target)) return -1;
            HIP_TRY(hipMemcpyAsync(h->hist_u0 + i * B * 2, h->u0, B * 2 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(h->hist_x0 + i * B * 8, h->x0, B * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(h->hist_st + i * B, h->status, B * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(h->hist_it + i * B, h->qp_iter, B * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
        }
    }
    HIP_TRY(hipGetLastError());
    // histories: stream-ordered copies; pinned destinations (ihm2mpc_host_alloc) do not block the host
    if (u0_hist) HIP_TRY(hipMemcpyAsync(u0_hist, h->hist_u0, n * B * 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (x0_hist) HIP_TRY(hipMemcpyAsync(x0_hist, h->hist_x0, n * B * 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (status_hist) HIP_TRY(hipMemcpyAsync(status_hist, h->hist_st, n * B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (qp_iter_hist) HIP_TRY(hipMemcpyAsync(qp_iter_hist, h->hist_it, n * B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    return 0;
}

// ---- Cartesian side of the ROS stack (N2 plants, N3 projection) ----
int ihm2mpc_set_track_geometry(ihm2mpc_handle *h, const double *X_ref, const double *Y_ref, const double *phi_ref)
{
    CHECK_H(h);
    if (!X_ref || !Y_ref || !phi_ref) return fail("null argument");
    const size_t n = (size_t)h->cfg.ntracks * h->cfg.nknots;
    if (upload_shared(h, X_ref, h->X_ref, n) || upload_shared(h, Y_ref, h->Y_ref, n) || upload_shared(h, phi_ref, h->phi_ref, n)) return -1;
    h->geometry_set = true;
    return 0;
}

static int check_cart(ihm2mpc_handle *h, int32_t model, int32_t M_sim, double dt_sim, int32_t n_steps)
{
    if (model != IHM2MPC_PLANT_KIN6 && model != IHM2MPC_PLANT_DYN6 && model != IHM2MPC_PLANT_ROS) return fail("unknown Cartesian plant %d", model);
    if (M_sim < 1 || n_steps < 1) return fail("M_sim and n_steps must be >= 1");
    if (!(dt_sim > 0.0)) return fail("dt_sim must be positive");
    (void)h;
    return 0;
}

int ihm2mpc_sim_step_cart(ihm2mpc_handle *h, int32_t model, int32_t M_sim, double dt_sim, int32_t n_steps, double v_dyn,
                          const double *x, const double *u, double *x_next)
{
    CHECK_H(h);
    if (!x || !u || !x_next) return fail("null argument");
    if (check_cart(h, model, M_sim, dt_sim, n_steps)) return -1;
    double *xs = h->scratch, *us = h->scratch + (size_t)h->B * 8, *xn = h->scratch + (size_t)h->B * 16;
    if (upload(h, x, xs, NX) || upload(h, u, us, NU)) return -1;
    ihm2_launch_sim_cart(h, model, M_sim, dt_sim, n_steps, v_dyn, xs, us, xn, h->stream);
    HIP_TRY(hipGetLastError());
    return download(h, xn, x_next, NX);
}

int ihm2mpc_sim_step_dyn10(ihm2mpc_handle *h, int32_t M_sim, const double *x, const double *u, double *x_next)
{
    CHECK_H(h);
    if (!h->tracks_set) return fail("ihm2mpc_set_tracks has not been called");
    if (!x || !u || !x_next) return fail("null argument");
    if (M_sim < 1) return fail("M_sim must be >= 1");
    const bool irk = h->cfg.sim_integrator_type != IHM2MPC_INTEG_ERK;
    if (!irk && rk4_unstable(h->cfg.dt, M_sim)) return fail("RK4 with M_sim = %d sub-steps of dt = %g is unstable on the torque lags (t_T = 1e-3 s): use M_sim >= %d", M_sim, h->cfg.dt, (int)ceil(h->cfg.dt / (2.78 * 1e-3)));
    if (!h->dyn10) { HIP_TRY(hipMalloc((void **)&h->dyn10, (size_t)h->B * 35 * sizeof(double))); }
    double *xs = h->dyn10, *us = h->dyn10 + (size_t)h->B * 15, *xn = h->dyn10 + (size_t)h->B * 20;
    if (upload(h, x, xs, 15) || upload(h, u, us, 5)) return -1;
    // the handle's plant integrator (python/main.py:395-400: IRK, GAUSS_RADAU_IIA): collocation steps of at most dt / M_sim, each
    // solved to convergence within IHM2MPC_DYN10_NEWTON_MAX Newton iterations or cut (kernels_dyn10.hip) -- usable from rest
    if (irk) ihm2_launch_sim_dyn10_irk(h, h->cfg.sim_integrator_type, M_sim, IHM2MPC_DYN10_NEWTON_MAX, xs, us, xn, h->stream);
    else ihm2_launch_sim_dyn10(h, M_sim, xs, us, xn, h->stream);
    HIP_TRY(hipGetLastError());
    return download(h, xn, x_next, 15);
}

int ihm2mpc_project(ihm2mpc_handle *h, const double *x_cart, double *s_guess, double s_tol, double *x_frenet)
{
    CHECK_H(h);
    if (!x_cart || !s_guess || !x_frenet) return fail("null argument");
    if (!h->tracks_set || !h->geometry_set) return fail("ihm2mpc_set_tracks / ihm2mpc_set_track_geometry have not been called");
    if (!(s_tol > 0.0)) return fail("s_tol must be positive");
    double *xs = h->scratch, *sg = h->scratch + (size_t)h->B * 8, *xf = h->scratch + (size_t)h->B * 9;
    if (upload(h, x_cart, xs, NX) || upload(h, s_guess, sg, 1)) return -1;
    ihm2_launch_project(h, s_tol, xs, sg, xf, h->stream);
    HIP_TRY(hipGetLastError());
    if (download(h, xf, x_frenet, NX)) return -1;
    return download(h, sg, s_guess, 1);
}

int ihm2mpc_set_cart_state(ihm2mpc_handle *h, const double *x_cart, const double *s_guess)
{
    CHECK_H(h);
    if (!x_cart || !s_guess) return fail("null argument");
    if (upload(h, x_cart, h->xc, NX)) return -1;
    return upload(h, s_guess, h->s_guess, 1);
}

int ihm2mpc_get_cart_state(ihm2mpc_handle *h, double *x_cart, double *s_guess)
{
    CHECK_H(h);
    if (x_cart && download(h, h->xc, x_cart, NX)) return -1;
    if (s_guess && download(h, h->s_guess, s_guess, 1)) return -1;
    return 0;
}

int ihm2mpc_sim_advance_cart(ihm2mpc_handle *h, int32_t model, int32_t M_sim, double dt_sim, int32_t n_steps, double v_dyn, double s_tol)
{
    CHECK_H(h);
    if (!h->tracks_set || !h->geometry_set) return fail("ihm2mpc_set_tracks / ihm2mpc_set_track_geometry have not been called");
    if (check_cart(h, model, M_sim, dt_sim, n_steps)) return -1;
    if (!(s_tol > 0.0)) return fail("s_tol must be positive");
    ihm2_launch_sim_cart(h, model, M_sim, dt_sim, n_steps, v_dyn, h->xc, h->u0, h->xc, h->stream);
    ihm2_launch_project(h, s_tol, h->xc, h->s_guess, h->x0, h->stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
