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

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

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
    return 0;
}

// host AoS (B, elems) -> device SoA
int upload(ihm2mpc_handle *h, const double *host, double *soa, int elems)
{
    const size_t n = (size_t)h->B * elems;
    if (n > h->stage_elems) return fail("staging buffer too small");
    memcpy(h->stage_h, host, n * sizeof(double));
    HIP_TRY(hipMemcpyAsync(h->stage_d, h->stage_h, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    ihm2_launch_aos_to_soa(h, h->stage_d, soa, elems);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));   // stage_h is reused by the next setter
    return 0;
}

int download(ihm2mpc_handle *h, const double *soa, double *host, int elems)
{
    const size_t n = (size_t)h->B * elems;
    if (n > h->stage_elems) return fail("staging buffer too small");
    ihm2_launch_soa_to_aos(h, soa, h->stage_d, elems);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->stage_h, h->stage_d, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    memcpy(host, h->stage_h, n * sizeof(double));
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

extern "C" {

const char *ihm2mpc_last_error(void) { return g_err.c_str(); }
const char *ihm2mpc_version(void) { return "ihm2mpc 0.1 (gfx950)"; }

int ihm2mpc_create(const ihm2mpc_config *cfg, ihm2mpc_handle **out)
{
    if (!cfg || !out) return fail("null argument");
    if (cfg->batch < 1) return fail("batch must be >= 1");
    if (cfg->N < 2 || cfg->N > IHM2MPC_NMAX) return fail("N must be in [2, %d]", IHM2MPC_NMAX);
    if (cfg->M < 1) return fail("M must be >= 1");
    if (cfg->model != IHM2MPC_MODEL_FKIN6) return fail("OCP model %d is not implemented (only fkin6)", cfg->model);
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
    h->Bp = (cfg->batch + 63) / 64 * 64;
    h->N = cfg->N;
    h->NS = cfg->N + 1;
    const size_t Bp = h->Bp, N = h->N, NS = h->NS;
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&h->ev[i]));
#define DA(p, n) if (dalloc(&h->p, (n))) return -1
    DA(s_ref, (size_t)cfg->ntracks * cfg->nknots); DA(kappa_ref, (size_t)cfg->ntracks * cfg->nknots);
    DA(track_id, Bp);
    DA(Hs, NS * 100); DA(Gy, NS * 120); DA(lbx, NS * 8); DA(ubx, NS * 8); DA(lbu, N * 2); DA(ubu, N * 2);
    DA(CD, N * 20); DA(lg, N * 2); DA(ug, N * 2);
    DA(x, NS * 8 * Bp); DA(u, N * 2 * Bp); DA(x0, 8 * Bp); DA(yref, N * 12 * Bp); DA(yref_e, 8 * Bp);
    DA(pi, NS * 8 * Bp); DA(lam, NS * 24 * Bp); DA(res, 4 * Bp); DA(status, Bp); DA(qp_iter, Bp); DA(u0, 2 * Bp);
    DA(A, N * 64 * Bp); DA(Bm, N * 16 * Bp); DA(bvec, N * 8 * Bp);
    DA(q_g, NS * 10 * Bp); DA(q_dl, NS * 12 * Bp); DA(q_du, NS * 12 * Bp); DA(q_z, NS * 10 * Bp); DA(q_pi, NS * 8 * Bp);
    DA(q_lam, NS * 24 * Bp); DA(q_t, NS * 24 * Bp); DA(q_gt, NS * 10 * Bp); DA(q_rb, N * 8 * Bp); DA(q_rd, NS * 24 * Bp);
    DA(q_dz, NS * 10 * Bp); DA(q_dpi, NS * 8 * Bp); DA(q_dlam, NS * 24 * Bp); DA(q_dt, NS * 24 * Bp);
    DA(q_dlam_a, NS * 24 * Bp); DA(q_dt_a, NS * 24 * Bp); DA(q_P, NS * 36 * Bp); DA(q_Gux, N * 16 * Bp);
    DA(q_Ginv, N * 3 * Bp); DA(q_p, NS * 8 * Bp); DA(q_kff, N * 2 * Bp);
#undef DA
    h->stage_elems = (size_t)h->B * (N * 64 > NS * 24 ? N * 64 : NS * 24);
    HIP_TRY(hipMalloc((void **)&h->stage_d, h->stage_elems * sizeof(double)));
    HIP_TRY(hipHostMalloc((void **)&h->stage_h, h->stage_elems * sizeof(double), hipHostMallocDefault));
    *out = h;
    return 0;
}

int ihm2mpc_free(ihm2mpc_handle *h)
{
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    void *ptrs[] = {h->s_ref, h->kappa_ref, h->track_id, h->Hs, h->Gy, h->lbx, h->ubx, h->lbu, h->ubu, h->CD, h->lg, h->ug,
                    h->x, h->u, h->x0, h->yref, h->yref_e, h->pi, h->lam, h->res, h->status, h->qp_iter, h->u0, h->A, h->Bm,
                    h->bvec, h->q_g, h->q_dl, h->q_du, h->q_z, h->q_pi, h->q_lam, h->q_t, h->q_gt, h->q_rb, h->q_rd, h->q_dz,
                    h->q_dpi, h->q_dlam, h->q_dt, h->q_dlam_a, h->q_dt_a, h->q_P, h->q_Gux, h->q_Ginv, h->q_p, h->q_kff,
                    h->stage_d};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (h->stage_h) (void)hipHostFree(h->stage_h);
    for (int i = 0; i < 4; i++) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
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
    if (upload_shared(h, Hs.data(), h->Hs, Hs.size()) || upload_shared(h, Gy.data(), h->Gy, Gy.size())) return -1;
    h->weights_set = true;
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
    if (upload_shared(h, lbx, h->lbx, (size_t)NS * 8) || upload_shared(h, ubx, h->ubx, (size_t)NS * 8) ||
        upload_shared(h, lbu, h->lbu, (size_t)N * 2) || upload_shared(h, ubu, h->ubu, (size_t)N * 2) ||
        upload_shared(h, CD.data(), h->CD, CD.size()) || upload_shared(h, lg, h->lg, (size_t)N * 2) ||
        upload_shared(h, ug, h->ug, (size_t)N * 2))
        return -1;
    h->bounds_set = true;
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
    else { ihm2_launch_fill(h, h->pi, h->NS * NX, 0.0); }
    if (lam) { if (upload(h, lam, h->lam, h->NS * NLAM)) return -1; }
    else { ihm2_launch_fill(h, h->lam, h->NS * NLAM, 0.0); }
    HIP_TRY(hipGetLastError());
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
    double *dst = fi.base + (size_t)stage * fi.per_stage * h->Bp + instance;
    HIP_TRY(hipMemcpy2DAsync(dst, (size_t)h->Bp * sizeof(double), value, sizeof(double), sizeof(double), n, hipMemcpyHostToDevice, h->stream));
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
    const double *src = fi.base + (size_t)stage * fi.per_stage * h->Bp + instance;
    HIP_TRY(hipMemcpy2DAsync(value, sizeof(double), src, (size_t)h->Bp * sizeof(double), sizeof(double), n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int ihm2mpc_init_guess(ihm2mpc_handle *h, double v_ref_scale)
{
    CHECK_H(h);
    if (ready(h)) return -1;
    ihm2_launch_init_guess(h, v_ref_scale);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ihm2mpc_prepare_step(ihm2mpc_handle *h, double s_target)
{
    CHECK_H(h);
    ihm2_launch_prepare(h, s_target);
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

int ihm2mpc_solve(ihm2mpc_handle *h, int32_t n_iter)
{
    CHECK_H(h);
    if (ready(h)) return -1;
    if (n_iter <= 0) n_iter = (h->cfg.nlp_solver_type == IHM2MPC_SQP) ? (h->cfg.nlp_solver_max_iter > 0 ? h->cfg.nlp_solver_max_iter : 1) : 1;
    HIP_TRY(hipEventRecord(h->ev[0], h->stream));
    for (int it = 0; it < n_iter; it++) {
        ihm2_launch_linearize(h);
        if (it == n_iter - 1) HIP_TRY(hipEventRecord(h->ev[1], h->stream));
        ihm2_launch_qp(h);
    }
    HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    HIP_TRY(hipGetLastError());
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
    if (A && download(h, h->A, A, h->N * 64)) return -1;
    if (Bm && download(h, h->Bm, Bm, h->N * 16)) return -1;
    if (b && download(h, h->bvec, b, h->N * 8)) return -1;
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
GETTER(x0, x0, NX)
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
int ihm2mpc_set_x0_device(ihm2mpc_handle *h, const void *dptr)
{
    CHECK_H(h);
    if (!dptr) return fail("null argument");
    ihm2_launch_aos_to_soa(h, (const double *)dptr, h->x0, NX);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ihm2mpc_get_u0_device(ihm2mpc_handle *h, void *dptr)
{
    CHECK_H(h);
    if (!dptr) return fail("null argument");
    ihm2_launch_soa_to_aos(h, h->u0, (double *)dptr, NU);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ihm2mpc_get_x_device(ihm2mpc_handle *h, void *dptr)
{
    CHECK_H(h);
    if (!dptr) return fail("null argument");
    ihm2_launch_soa_to_aos(h, h->x, (double *)dptr, h->NS * NX);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ihm2mpc_get_u_device(ihm2mpc_handle *h, void *dptr)
{
    CHECK_H(h);
    if (!dptr) return fail("null argument");
    ihm2_launch_soa_to_aos(h, h->u, (double *)dptr, h->N * NU);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ihm2mpc_get_status_device(ihm2mpc_handle *h, void *dptr)
{
    CHECK_H(h);
    if (!dptr) return fail("null argument");
    HIP_TRY(hipMemcpyAsync(dptr, h->status, (size_t)h->B * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

// ---- plant ----
int ihm2mpc_sim_step(ihm2mpc_handle *h, int32_t model, int32_t M_sim, const double *x, const double *u, double *x_next)
{
    CHECK_H(h);
    if (!h->tracks_set) return fail("ihm2mpc_set_tracks has not been called");
    if (!x || !u || !x_next) return fail("null argument");
    if (M_sim < 1) return fail("M_sim must be >= 1");
    if (model < -1 || model > IHM2MPC_MODEL_FDYN6) return fail("unknown plant model %d", model);
    // scratch: the QP step buffers are free between solves
    double *xs = h->q_dz, *us = h->q_dpi, *xn = h->q_gt;
    if (upload(h, x, xs, NX) || upload(h, u, us, NU)) return -1;
    ihm2_launch_sim(h, model, M_sim, xs, us, xn);
    HIP_TRY(hipGetLastError());
    return download(h, xn, x_next, NX);
}

int ihm2mpc_sim_advance(ihm2mpc_handle *h, int32_t model, int32_t M_sim)
{
    CHECK_H(h);
    if (!h->tracks_set) return fail("ihm2mpc_set_tracks has not been called");
    if (M_sim < 1) return fail("M_sim must be >= 1");
    if (model < -1 || model > IHM2MPC_MODEL_FDYN6) return fail("unknown plant model %d", model);
    ihm2_launch_sim(h, model, M_sim, h->x0, h->u0, h->x0);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
