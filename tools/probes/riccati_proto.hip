// Stand-alone check + timing of ihm2_amd/csrc/riccati_mfma.hpp against a plain CPU Riccati recursion on random data.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I ihm2_amd/csrc tools/probes/riccati_proto.hip -o /tmp/riccati_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "riccati_mfma.hpp"
using namespace ihm2;
#ifndef PATHV
#define PATHV 0
#endif
#ifndef RD
#define RD 4
#endif
#ifndef UNIV
#define UNIV 1
#endif
constexpr int NCK = PATHV ? 14 : 12;
constexpr int REC = 96;

struct Off { int z, pi, gam, hc, gt, rb, pv, hv, Kl, Ginv, kff, dz, tile, total; };
static Off offsets(int N)
{
    const int NS = N + 1; Off o; int p = 0;
    o.z = p; p += NS * 10; o.pi = p; p += NS * 8; o.gam = p; p += NS * NCK; o.hc = p; p += NS * 2; o.gt = p; p += NS * 10;
    o.rb = p; p += N * 8; o.pv = p; p += NS * 8; o.hv = p; p += N * 8; o.Kl = p; p += N * 16; o.Ginv = p; p += N * 8; o.kff = p; p += N * 4; o.dz = p; p += NS * 10; o.tile = p; p += 136;
    o.total = p; return o;
}

__global__ __launch_bounds__(64) void k_proto(int N, Off o, double *lin_all, const double *Hs, const double *CD, const double *lds_in, double *lds_out,
                                              double *Pg, double *Mg, long long *cyc, int reps)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x, b = blockIdx.x;
    double *lin = lin_all + (size_t)b * N * REC;
    for (int e = lane; e < o.total; e += 64) sm[e] = lds_in[e];
    __syncthreads();
    RicLds L; L.gam = o.gam; L.hc = o.hc; L.gt = o.gt; L.pv = o.pv; L.hv = o.hv; L.Kl = o.Kl; L.Ginv = o.Ginv; L.kff = o.kff; L.dz = o.dz; L.tile = o.tile;
    long long t0 = 0, tr = 0;
    for (int r = 0; r < reps; r++) {
        long long tq = __builtin_readcyclecounter();
        if (r > 0) { for (int e = lane; e < (N + 1) * 10; e += 64) sm[o.gt + e] = lds_in[o.gt + e]; __syncthreads(); }
        dyn_residual<4>(N, lane, lin, sm + o.z, sm + o.pi, sm + o.gt, sm + o.rb, REC);
        __syncthreads();
        if (r == 0) t0 = __builtin_readcyclecounter(), tr += t0 - tq;
        riccati_sweep_mfma<NCK, PATHV != 0, UNIV != 0, RD>(N, lane, lin, Hs, CD, L, Pg + (size_t)b * (N + 1) * 64, Mg + (size_t)b * N * 64, REC, true, true);
        __syncthreads();
    }
    long long t1 = __builtin_readcyclecounter();
    if (b == 0) for (int e = lane; e < o.total; e += 64) lds_out[e] = sm[e];
    if (lane == 0) { cyc[b] = t1 - t0; cyc[gridDim.x + b] = tr; }
}

static double rnd() { return 2.0 * rand() / (double)RAND_MAX - 1.0; }
static double maxrel(const double *a, const double *b, int n, const char *name)
{
    double e = 0, sc = 0;
    for (int i = 0; i < n; i++) { sc = fmax(sc, fabs(b[i])); }
    for (int i = 0; i < n; i++) e = fmax(e, fabs(a[i] - b[i]));
    printf("  %-6s max abs err %.3e  (scale %.3e)  rel %.3e\n", name, e, sc, e / fmax(sc, 1e-300));
    return e / fmax(sc, 1e-300);
}

int main(int argc, char **argv)
{
    const int N = (argc > 1) ? atoi(argv[1]) : 40, NS = N + 1;
    const int NB = (argc > 2) ? atoi(argv[2]) : 1024;
    srand(7);
    Off o = offsets(N);
    std::vector<double> lin(N * REC), Hs(NS * 100), CD(N * 20), lds(o.total, 0.0);
    for (int k = 0; k < NS; k++) {
        double R[100];
        for (int i = 0; i < 100; i++) R[i] = 0.3 * rnd();
        for (int i = 0; i < 10; i++) for (int j = 0; j < 10; j++) {
            double s = (i == j) ? 1.0 + i : 0.0;
            for (int l = 0; l < 10; l++) s += R[i * 10 + l] * R[j * 10 + l];
            Hs[k * 100 + i * 10 + j] = s;
        }
        if (UNIV && k > 0 && k < N) for (int i = 0; i < 100; i++) Hs[k * 100 + i] = Hs[i];
    }
    for (int k = 0; k < N; k++) for (int i = 0; i < 20; i++) CD[k * 20 + i] = (UNIV && k > 0) ? CD[i] : rnd();
    for (int k = 0; k < N; k++) {
        for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) lin[k * REC + i * 8 + j] = (i == j ? 1.0 : 0.0) + 0.2 * rnd();
        for (int i = 0; i < 16; i++) lin[k * REC + 64 + i] = rnd();
        for (int i = 0; i < 8; i++) lin[k * REC + 80 + i] = 0.1 * rnd();
        for (int i = 0; i < 8; i++) lin[k * REC + 88 + i] = 1e300;      // the rb slot must be rewritten
    }
    for (int e = 0; e < NS * 10; e++) { lds[o.z + e] = rnd(); lds[o.gt + e] = rnd(); }
    for (int e = 0; e < NS * 8; e++) lds[o.pi + e] = rnd();
    for (int e = 0; e < NS * NCK; e++) lds[o.gam + e] = (rand() % 3 == 0) ? 0.0 : exp(6.0 * rnd());
    for (int e = 0; e < NS * 2; e++) lds[o.hc + e] = rnd();

    // ---- CPU reference ----
    std::vector<double> P(NS * 64), M(N * 64), Kl(N * 16), Gi(N * 8, 0.0), kff(N * 4, 0.0), pv(NS * 8), hv(N * 8), rb(N * 8), cz(NS * 10, 0.0), gt(NS * 10);
    auto Ht = [&](int k, int i, int j) {
        double v = Hs[k * 100 + i * 10 + j];
        const double *gam = &lds[o.gam + k * NCK];
        if (i == j) v += gam[i];
        if (k < N) v += gam[10] * CD[k * 20 + i] * CD[k * 20 + j] + gam[11] * CD[k * 20 + 10 + i] * CD[k * 20 + 10 + j];
        if (PATHV && (i == 1 || i == 2) && (j == 1 || j == 2)) {
            const double g12 = gam[12], g13 = gam[13], a0 = lds[o.hc + k * 2], a1 = lds[o.hc + k * 2 + 1];
            v += (i == 1 && j == 1) ? g12 + g13 : (i == 2 && j == 2) ? g12 * a0 * a0 + g13 * a1 * a1 : g12 * a0 - g13 * a1;
        }
        return v;
    };
    for (int e = 0; e < NS * 10; e++) gt[e] = lds[o.gt + e];
    for (int k = 0; k < N; k++) {
        const double *A = &lin[k * REC], *Bm = A + 64, *b = A + 80;
        for (int i = 0; i < 8; i++) {
            double s = b[i] - lds[o.z + (k + 1) * 10 + i];
            for (int l = 0; l < 8; l++) s += A[i * 8 + l] * lds[o.z + k * 10 + l];
            s += Bm[i * 2] * lds[o.z + k * 10 + 8] + Bm[i * 2 + 1] * lds[o.z + k * 10 + 9];
            rb[k * 8 + i] = s;
        }
        for (int jz = 0; jz < 10; jz++) {
            double s = 0;
            for (int l = 0; l < 8; l++) s += ((jz < 8) ? A[l * 8 + jz] : Bm[l * 2 + jz - 8]) * lds[o.pi + (k + 1) * 8 + l];
            gt[k * 10 + jz] += s;
        }
    }
    for (int i = 0; i < 8; i++) { for (int j = 0; j < 8; j++) P[N * 64 + i * 8 + j] = Ht(N, i, j); pv[N * 8 + i] = gt[N * 10 + i]; }
    for (int k = N - 1; k >= 0; k--) {
        const double *A = &lin[k * REC], *Bm = A + 64;
        const double *Pn = &P[(k + 1) * 64];
        auto AB = [&](int l, int jz) { return (jz < 8) ? A[l * 8 + jz] : Bm[l * 2 + jz - 8]; };
        double W[8][10], G[10][10], gv[10];
        for (int i = 0; i < 8; i++) for (int jz = 0; jz < 10; jz++) { double s = 0; for (int l = 0; l < 8; l++) s += Pn[i * 8 + l] * AB(l, jz); W[i][jz] = s; }
        for (int i = 0; i < 10; i++) for (int jz = 0; jz < 10; jz++) { double s = Ht(k, i, jz); for (int l = 0; l < 8; l++) s += AB(l, i) * W[l][jz]; G[i][jz] = s; }
        for (int i = 0; i < 8; i++) { double s = 0; for (int l = 0; l < 8; l++) s += Pn[i * 8 + l] * rb[k * 8 + l]; hv[k * 8 + i] = s; }
        for (int jz = 0; jz < 10; jz++) { double s = gt[k * 10 + jz]; for (int l = 0; l < 8; l++) s += AB(l, jz) * (hv[k * 8 + l] + pv[(k + 1) * 8 + l]); gv[jz] = s; }
        const double det = G[8][8] * G[9][9] - G[8][9] * G[8][9];
        const double Gi0 = G[9][9] / det, Gi1 = -G[8][9] / det, Gi2 = G[8][8] / det;
        Gi[k * 8] = Gi0; Gi[k * 8 + 1] = Gi1; Gi[k * 8 + 2] = Gi2; Gi[k * 8 + 3] = Gi1;
        for (int jz = 0; jz < 8; jz++) { Kl[k * 16 + jz] = Gi0 * G[8][jz] + Gi1 * G[9][jz]; Kl[k * 16 + 8 + jz] = Gi1 * G[8][jz] + Gi2 * G[9][jz]; }
        kff[k * 4] = Gi0 * gv[8] + Gi1 * gv[9]; kff[k * 4 + 1] = Gi1 * gv[8] + Gi2 * gv[9];
        for (int i = 0; i < 8; i++) {
            for (int jz = 0; jz < 8; jz++) {
                P[k * 64 + i * 8 + jz] = G[i][jz] - G[i][8] * Kl[k * 16 + jz] - G[i][9] * Kl[k * 16 + 8 + jz];
                M[k * 64 + i * 8 + jz] = A[i * 8 + jz] - Bm[i * 2] * Kl[k * 16 + jz] - Bm[i * 2 + 1] * Kl[k * 16 + 8 + jz];
            }
            pv[k * 8 + i] = gv[i] - G[i][8] * kff[k * 4] - G[i][9] * kff[k * 4 + 1];
            cz[(k + 1) * 10 + i] = rb[k * 8 + i] - Bm[i * 2] * kff[k * 4] - Bm[i * 2 + 1] * kff[k * 4 + 1];
        }
    }

    // ---- GPU ----
    double *dlin, *dHs, *dCD, *dli, *dlo, *dP, *dM; long long *dc;
    hipMalloc(&dlin, (size_t)NB * lin.size() * 8); hipMalloc(&dHs, Hs.size() * 8); hipMalloc(&dCD, CD.size() * 8); hipMalloc(&dli, o.total * 8); hipMalloc(&dlo, o.total * 8);
    hipMalloc(&dP, (size_t)NB * NS * 64 * 8); hipMalloc(&dM, (size_t)NB * N * 64 * 8); hipMalloc(&dc, 2 * NB * 8);
    for (int b = 0; b < NB; b++) hipMemcpy(dlin + (size_t)b * lin.size(), lin.data(), lin.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dHs, Hs.data(), Hs.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dCD, CD.data(), CD.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dli, lds.data(), o.total * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_proto, hipFuncAttributeMaxDynamicSharedMemorySize, o.total * 8);
    hipLaunchKernelGGL(k_proto, dim3(1), dim3(64), o.total * 8, 0, N, o, dlin, dHs, dCD, dli, dlo, dP, dM, dc, 1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<double> out(o.total), gPr(NS * 64), gMr(N * 64), gP(NS * 64), gM(N * 64);
    hipMemcpy(out.data(), dlo, o.total * 8, hipMemcpyDeviceToHost); hipMemcpy(gPr.data(), dP, NS * 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(gMr.data(), dM, N * 64 * 8, hipMemcpyDeviceToHost);
    for (int k = 0; k < NS; k++) for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { gP[k * 64 + i * 8 + j] = gPr[k * 64 + RIC_IDX(i, j)]; if (k < N) gM[k * 64 + i * 8 + j] = gMr[k * 64 + RIC_IDX(i, j)]; }
    printf("N = %d PATH = %d UNI = %d ring %d, LDS %d B\n", N, PATHV, UNIV, RD, o.total * 8);
    double w = 0;
    w = fmax(w, maxrel(gP.data(), P.data(), NS * 64, "P"));
    w = fmax(w, maxrel(gM.data(), M.data(), N * 64, "M"));
    w = fmax(w, maxrel(&out[o.Kl], Kl.data(), N * 16, "K"));
    w = fmax(w, maxrel(&out[o.Ginv], Gi.data(), N * 8, "Ginv"));
    std::vector<double> gk, ck; for (int k = 0; k < N; k++) for (int q = 0; q < 2; q++) { gk.push_back(out[o.kff + k * 4 + q]); ck.push_back(kff[k * 4 + q]); }
    w = fmax(w, maxrel(gk.data(), ck.data(), N * 2, "kff"));
    w = fmax(w, maxrel(&out[o.pv], pv.data(), NS * 8, "p"));
    w = fmax(w, maxrel(&out[o.hv], hv.data(), N * 8, "hv"));
    w = fmax(w, maxrel(&out[o.rb], rb.data(), N * 8, "rb"));
    std::vector<double> gc, cc; for (int k = 1; k <= N; k++) for (int i = 0; i < 8; i++) { gc.push_back(out[o.dz + k * 10 + i]); cc.push_back(cz[k * 10 + i]); }
    w = fmax(w, maxrel(gc.data(), cc.data(), N * 8, "c"));
    w = fmax(w, maxrel(&out[o.gt], gt.data(), N * 10, "gt"));
    printf("worst relative deviation %.3e %s\n", w, (w < 1e-9) ? "OK" : "FAIL");
#ifdef RIC_STAMPS
    { long long T[8]; hipMemcpyFromSymbol(T, HIP_SYMBOL(ric_dbg), sizeof T); const char *nm[] = {"W (2 mfma)", "G (2 mfma)", "K", "S (mfma)", "ring + prepare", "stores", "loop", "-"};
      for (int q = 0; q < 7; q++) printf("  section %-16s %8.1f cycles/stage\n", nm[q], (double)T[q] / N); }
#endif
    const int reps = 20;
    for (int nb : {1, NB}) {
        hipLaunchKernelGGL(k_proto, dim3(nb), dim3(64), o.total * 8, 0, N, o, dlin, dHs, dCD, dli, dlo, dP, dM, dc, reps);
        hipDeviceSynchronize();
        std::vector<long long> c(2 * nb); hipMemcpy(c.data(), dc, 2 * nb * 8, hipMemcpyDeviceToHost);
        double mean = 0, mr = 0; for (int q = 0; q < nb; q++) { mean += c[q]; mr += c[nb + q]; } mean /= nb; mr /= nb;
        printf("blocks %5d: %.0f cycles per (residual phase + sweep), %.1f per stage; first residual phase %.0f cycles\n", nb, mean / reps, mean / reps / N, mr);
    }
    return 0;
}
