// Stand-alone check + timing of ihm2_amd/csrc/riccati_blk.hpp (backward Riccati sweep on blocks of two shooting intervals) against a plain CPU
// recursion on random data, and against the stage-wise matrix-core sweep it is meant to replace (cycles per two stages).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I ihm2_amd/csrc -I tools/probes tools/probes/riccati_blk_proto.hip -o tools/probes/bin/blk_proto   (add -DRIC_STAMPS for the per-section stamps)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "riccati_blk.hpp"
using namespace ihm2;
#ifndef RD
#define RD 2
#endif
constexpr int NCK = 12;
constexpr int REC = 96;

struct Off { int gam, gt, pv, hv, Kl, Ginv, kff, dz, total; };
static Off offsets(int N)
{
    const int NS = N + 1; Off o; int p = 0;
    o.gam = p; p += NS * NCK; o.gt = p; p += NS * 10; o.pv = p; p += NS * 8; o.hv = p; p += N * 4; o.Kl = p; p += N * 16; o.Ginv = p; p += N * 8;
    o.kff = p; p += N * 2; o.dz = p; p += NS * 10;
    o.total = p; return o;
}

__global__ __launch_bounds__(64) void k_proto(int N, Off o, const double *brec_all, const double *Hs, const double *CD, const double *wd, const double *lds_in,
                                              double *lds_out, double *Pg, double *Mg, long long *cyc, int reps)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x, b = blockIdx.x;
    const double *brec = brec_all + (size_t)b * (N / 2) * BREC;
    for (int e = lane; e < o.total; e += 64) sm[e] = lds_in[e];
    __syncthreads();
    BlkLds L; L.gam = o.gam; L.gt = o.gt; L.pv = o.pv; L.hv = o.hv; L.Kl = o.Kl; L.Ginv = o.Ginv; L.kff = o.kff; L.dz = o.dz;
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
        riccati_sweep_blk2<NCK, RD>(N, lane, brec, Hs, CD, wd, L, Pg + (size_t)b * (N + 1) * 64, Mg + (size_t)b * (N / 2) * 64, true);
        __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (b == 0) for (int e = lane; e < o.total; e += 64) lds_out[e] = sm[e];
    if (lane == 0) cyc[b] = t1 - t0;
}

static double rnd() { return 2.0 * rand() / (double)RAND_MAX - 1.0; }
static double maxrel(const double *a, const double *b, int n, const char *name)
{
    double e = 0, sc = 0;
    for (int i = 0; i < n; i++) sc = fmax(sc, fabs(b[i]));
    for (int i = 0; i < n; i++) e = fmax(e, fabs(a[i] - b[i]));
    printf("  %-6s max abs err %.3e  (scale %.3e)  rel %.3e\n", name, e, sc, e / fmax(sc, 1e-300));
    return e / fmax(sc, 1e-300);
}

int main(int argc, char **argv)
{
    const int N = (argc > 1) ? atoi(argv[1]) : 40, NS = N + 1, NBk = N / 2;
    const int NB = (argc > 2) ? atoi(argv[2]) : 1024;
    srand(7);
    Off o = offsets(N);
    // the reference's cost structure: y = [x; u; x6 - u0; x7 - u1] with a diagonal weight matrix (python/mpc.py:49-64), rate rows u - x[6:8]
    std::vector<double> wd(12), Hs(NS * 100, 0.0), CD(N * 20, 0.0), lds(o.total, 0.0), lin(N * REC);
    for (int r = 0; r < 12; r++) wd[r] = 0.05 * (1.0 + 3.0 * fabs(rnd()));
    double V[12][10] = {};
    for (int r = 0; r < 10; r++) V[r][r] = 1.0;
    V[10][6] = 1.0; V[10][8] = -1.0; V[11][7] = 1.0; V[11][9] = -1.0;
    for (int k = 0; k < N; k++) for (int i = 0; i < 10; i++) for (int j = 0; j < 10; j++) { double s = 0; for (int r = 0; r < 12; r++) s += wd[r] * V[r][i] * V[r][j]; Hs[k * 100 + i * 10 + j] = s; }
    for (int i = 0; i < 8; i++) Hs[N * 100 + i * 10 + i] = 1.0 + 10.0 * fabs(rnd());
    for (int k = 0; k < N; k++) { CD[k * 20 + 6] = -1.0; CD[k * 20 + 8] = 1.0; CD[k * 20 + 10 + 7] = -1.0; CD[k * 20 + 10 + 9] = 1.0; }
    for (int k = 0; k < N; k++) {
        for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) lin[k * REC + i * 8 + j] = (i == j ? 0.8 : 0.0) + 0.2 * rnd();
        for (int i = 0; i < 16; i++) lin[k * REC + 64 + i] = rnd();
        for (int i = 0; i < 8; i++) lin[k * REC + 88 + i] = 0.1 * rnd();         // rb
    }
    for (int e = 0; e < NS * 10; e++) lds[o.gt + e] = rnd();
    for (int e = 0; e < NS * NCK; e++) lds[o.gam + e] = (rand() % 3 == 0) ? 0.0 : exp(6.0 * rnd());
    for (int i = 8; i < NCK; i++) lds[o.gam + N * NCK + i] = 0.0;

    // ---- block records (what the condensing phase of the solver writes) ----
    std::vector<double> brec((size_t)NBk * BREC, 0.0);
    for (int m = 0; m < NBk; m++) {
        const double *Aa = &lin[(2 * m) * REC], *Ba = Aa + 64, *ra = Aa + 88, *Ab = &lin[(2 * m + 1) * REC], *Bb = Ab + 64, *rbb = Ab + 88;
        double *psi = &brec[(size_t)m * BREC + BREC_PSI], *yt = &brec[(size_t)m * BREC + BREC_Y];
        for (int i = 0; i < 8; i++) {
            for (int jz = 0; jz < 8; jz++) { double s = 0; for (int l = 0; l < 8; l++) s += Ab[i * 8 + l] * Aa[l * 8 + jz]; psi[i * 16 + jz] = s; }
            for (int c = 0; c < 2; c++) { double s = 0; for (int l = 0; l < 8; l++) s += Ab[i * 8 + l] * Ba[l * 2 + c]; psi[i * 16 + 8 + c] = s; psi[i * 16 + 10 + c] = Bb[i * 2 + c]; }
            { double s = rbb[i]; for (int l = 0; l < 8; l++) s += Ab[i * 8 + l] * ra[l]; psi[i * 16 + BREC_RT] = s; }
            for (int jz = 0; jz < 8; jz++) yt[i * 16 + jz] = Aa[i * 8 + jz];
            yt[i * 16 + 8] = Ba[i * 2]; yt[i * 16 + 9] = Ba[i * 2 + 1]; yt[i * 16 + BREC_RT] = ra[i];
        }
        for (int c = 0; c < 2; c++) { for (int jz = 0; jz < 16; jz++) yt[(8 + c) * 16 + jz] = yt[(6 + c) * 16 + jz]; yt[(8 + c) * 16 + 10 + c] = -1.0; }
    }

    // ---- CPU reference: the block recursion of tools/probes/block2_newton_model.py with dense stage Hessians ----
    auto Ht = [&](int k, int i, int jz) {
        double v = Hs[k * 100 + i * 10 + jz];
        const double *gam = &lds[o.gam + k * NCK];
        if (i == jz) v += gam[i];
        if (k < N) v += gam[10] * CD[k * 20 + i] * CD[k * 20 + jz] + gam[11] * CD[k * 20 + 10 + i] * CD[k * 20 + 10 + jz];
        return v;
    };
    std::vector<double> P(NS * 64, 0.0), Mt(NBk * 64), Kl(NBk * 32), Gi(NBk * 16), kff(NBk * 4), pv(NS * 8, 0.0), hv(NBk * 8), ct(NS * 10, 0.0);
    for (int i = 0; i < 8; i++) { for (int jz = 0; jz < 8; jz++) P[N * 64 + i * 8 + jz] = Ht(N, i, jz); pv[N * 8 + i] = lds[o.gt + N * 10 + i]; }
    for (int m = NBk - 1; m >= 0; m--) {
        const int a = 2 * m, b = a + 1;
        const double *Aa = &lin[a * REC], *Ba = Aa + 64, *ra = Aa + 88;
        const double *psi = &brec[(size_t)m * BREC];
        auto Y = [&](int l, int t) { return (t < 8) ? Aa[l * 8 + t] : Ba[l * 2 + t - 8]; };
        auto PS = [&](int l, int t) { return psi[l * 16 + t]; };        // t < 12
        double Hb[12][12] = {}, gb[12] = {};
        for (int i = 0; i < 10; i++) for (int jz = 0; jz < 10; jz++) { double s = Ht(a, i, jz); for (int l = 0; l < 8; l++) for (int q = 0; q < 8; q++) s += Y(l, i) * Ht(b, l, q) * Y(q, jz); Hb[i][jz] = s; }
        for (int i = 0; i < 10; i++) for (int c = 0; c < 2; c++) { double s = 0; for (int l = 0; l < 8; l++) s += Y(l, i) * Ht(b, l, 8 + c); Hb[i][10 + c] = Hb[10 + c][i] = s; }
        for (int c = 0; c < 2; c++) for (int e = 0; e < 2; e++) Hb[10 + c][10 + e] = Ht(b, 8 + c, 8 + e);
        for (int i = 0; i < 10; i++) { double s = lds[o.gt + a * 10 + i]; for (int l = 0; l < 8; l++) { double h = lds[o.gt + b * 10 + l]; for (int q = 0; q < 8; q++) h += Ht(b, l, q) * ra[q]; s += Y(l, i) * h; } gb[i] = s; }
        for (int c = 0; c < 2; c++) { double s = lds[o.gt + b * 10 + 8 + c]; for (int q = 0; q < 8; q++) s += Ht(b, 8 + c, q) * ra[q]; gb[10 + c] = s; }
        const double *Pn = &P[(a + 2) * 64];
        double W[8][12], G[12][12], gv[12], hvv[8];
        for (int i = 0; i < 8; i++) for (int t = 0; t < 12; t++) { double s = 0; for (int l = 0; l < 8; l++) s += Pn[i * 8 + l] * PS(l, t); W[i][t] = s; }
        for (int i = 0; i < 12; i++) for (int t = 0; t < 12; t++) { double s = Hb[i][t]; for (int l = 0; l < 8; l++) s += PS(l, i) * W[l][t]; G[i][t] = s; }
        for (int i = 0; i < 8; i++) { double s = 0; for (int l = 0; l < 8; l++) s += Pn[i * 8 + l] * psi[l * 16 + BREC_RT]; hvv[i] = s; hv[m * 8 + i] = s; }
        for (int t = 0; t < 12; t++) { double s = gb[t]; for (int l = 0; l < 8; l++) s += PS(l, t) * (hvv[l] + pv[(a + 2) * 8 + l]); gv[t] = s; }
        // Guu^-1 by Gauss-Jordan
        double Mx[4][8];
        for (int i = 0; i < 4; i++) for (int t = 0; t < 4; t++) { Mx[i][t] = G[8 + i][8 + t]; Mx[i][4 + t] = (i == t) ? 1.0 : 0.0; }
        for (int c = 0; c < 4; c++) { const double pvt = Mx[c][c]; for (int t = 0; t < 8; t++) Mx[c][t] /= pvt; for (int i = 0; i < 4; i++) if (i != c) { const double f = Mx[i][c]; for (int t = 0; t < 8; t++) Mx[i][t] -= f * Mx[c][t]; } }
        for (int i = 0; i < 4; i++) for (int t = 0; t < 4; t++) Gi[m * 16 + i * 4 + t] = Mx[i][4 + t];
        for (int i = 0; i < 4; i++) {
            for (int t = 0; t < 8; t++) { double s = 0; for (int l = 0; l < 4; l++) s += Gi[m * 16 + i * 4 + l] * G[8 + l][t]; Kl[m * 32 + i * 8 + t] = s; }
            double s = 0; for (int l = 0; l < 4; l++) s += Gi[m * 16 + i * 4 + l] * gv[8 + l]; kff[m * 4 + i] = s;
        }
        for (int i = 0; i < 8; i++) {
            for (int t = 0; t < 8; t++) {
                double sp = G[i][t], sm_ = PS(i, t);
                for (int l = 0; l < 4; l++) { sp -= G[i][8 + l] * Kl[m * 32 + l * 8 + t]; sm_ -= PS(i, 8 + l) * Kl[m * 32 + l * 8 + t]; }
                P[a * 64 + i * 8 + t] = sp; Mt[m * 64 + i * 8 + t] = sm_;
            }
            double sp = gv[i], sc = psi[i * 16 + BREC_RT];
            for (int l = 0; l < 4; l++) { sp -= G[i][8 + l] * kff[m * 4 + l]; sc -= PS(i, 8 + l) * kff[m * 4 + l]; }
            pv[a * 8 + i] = sp; ct[(a + 2) * 10 + i] = sc;
        }
    }

    // ---- GPU ----
    double *dbrec, *dHs, *dCD, *dwd, *dli, *dlo, *dP, *dM; long long *dc;
    hipMalloc(&dbrec, (size_t)NB * brec.size() * 8); hipMalloc(&dHs, Hs.size() * 8); hipMalloc(&dCD, CD.size() * 8); hipMalloc(&dwd, 12 * 8); hipMalloc(&dli, o.total * 8); hipMalloc(&dlo, o.total * 8);
    hipMalloc(&dP, (size_t)NB * NS * 64 * 8); hipMalloc(&dM, (size_t)NB * NBk * 64 * 8); hipMalloc(&dc, NB * 8);
    hipMemset(dP, 0, (size_t)NB * NS * 64 * 8);
    for (int b = 0; b < NB; b++) hipMemcpy(dbrec + (size_t)b * brec.size(), brec.data(), brec.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dHs, Hs.data(), Hs.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dCD, CD.data(), CD.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dwd, wd.data(), 12 * 8, hipMemcpyHostToDevice); hipMemcpy(dli, lds.data(), o.total * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_proto, hipFuncAttributeMaxDynamicSharedMemorySize, o.total * 8);
    hipLaunchKernelGGL(k_proto, dim3(1), dim3(64), o.total * 8, 0, N, o, dbrec, dHs, dCD, dwd, dli, dlo, dP, dM, dc, 1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<double> out(o.total), gPr(NS * 64), gMr(NBk * 64), gP(NS * 64, 0.0), gM(NBk * 64);
    hipMemcpy(out.data(), dlo, o.total * 8, hipMemcpyDeviceToHost); hipMemcpy(gPr.data(), dP, NS * 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(gMr.data(), dM, NBk * 64 * 8, hipMemcpyDeviceToHost);
    for (int k = 0; k < NS; k += 2) for (int i = 0; i < 8; i++) for (int t = 0; t < 8; t++) gP[k * 64 + i * 8 + t] = gPr[k * 64 + RIC_IDX(i, t)];
    for (int m = 0; m < NBk; m++) for (int i = 0; i < 8; i++) for (int t = 0; t < 8; t++) gM[m * 64 + i * 8 + t] = gMr[m * 64 + RIC_IDX(i, t)];
    printf("N = %d (blocks of two stages: %d block stages), ring %d, LDS %d B\n", N, NBk, RD, o.total * 8);
    double w = 0;
    w = fmax(w, maxrel(gP.data(), P.data(), NS * 64, "P"));
    w = fmax(w, maxrel(gM.data(), Mt.data(), NBk * 64, "Mt"));
    w = fmax(w, maxrel(&out[o.Kl], Kl.data(), NBk * 32, "K"));
    w = fmax(w, maxrel(&out[o.Ginv], Gi.data(), NBk * 16, "Ginv"));
    w = fmax(w, maxrel(&out[o.kff], kff.data(), NBk * 4, "kff"));
    std::vector<double> gp, cp; for (int k = 0; k < NS; k += 2) for (int i = 0; i < 8; i++) { gp.push_back(out[o.pv + k * 8 + i]); cp.push_back(pv[k * 8 + i]); }
    w = fmax(w, maxrel(gp.data(), cp.data(), (int)gp.size(), "p"));
    w = fmax(w, maxrel(&out[o.hv], hv.data(), NBk * 8, "hv"));
    std::vector<double> gc, cc; for (int k = 2; k <= N; k += 2) for (int i = 0; i < 8; i++) { gc.push_back(out[o.dz + k * 10 + i]); cc.push_back(ct[k * 10 + i]); }
    w = fmax(w, maxrel(gc.data(), cc.data(), (int)gc.size(), "ct"));
    printf("worst relative deviation %.3e %s\n", w, (w < 1e-9) ? "OK" : "FAIL");
#ifdef RIC_STAMPS
    { long long T[8]; hipMemcpyFromSymbol(T, HIP_SYMBOL(ric_dbg), sizeof T); const char *nm[] = {"tile (3 mfma)", "W, G (4 mfma)", "Guu^-1, K (mfma)", "S (mfma)", "ring + prepare", "stores", "loop", "-"};
      for (int q = 0; q < 7; q++) printf("  section %-18s %8.1f cycles/block\n", nm[q], (double)T[q] / NBk); }
#endif
    const int reps = 20;
    for (int nb : {1, NB}) {
        hipLaunchKernelGGL(k_proto, dim3(nb), dim3(64), o.total * 8, 0, N, o, dbrec, dHs, dCD, dwd, dli, dlo, dP, dM, dc, reps);
        hipDeviceSynchronize();
        std::vector<long long> c(nb); hipMemcpy(c.data(), dc, nb * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (int q = 0; q < nb; q++) mean += c[q]; mean /= nb;
        printf("blocks %5d: %.0f cycles per sweep, %.1f per block stage (= two shooting intervals)\n", nb, mean / reps, mean / reps / NBk);
    }
    return 0;
}
