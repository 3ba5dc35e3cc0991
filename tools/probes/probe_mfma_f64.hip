// Probe of v_mfma_f64_16x16x4_f64 on gfx950: operand/result lane layout and latencies of the primitives a register-resident
// Riccati stage would chain.  Standalone (hipcc probe.hip -o probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__global__ void k_layout(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = MFMA(a[l], b[l], acc);
    for (int r = 0; r < 4; r++) d[l * 4 + r] = acc[r];
}

template <int DPP>
__device__ __forceinline__ double dpp_mov(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, DPP, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, DPP, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swap16(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto a16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b16[0], a16[0]) + __hiloint2double(b16[1], a16[1]);
}
__device__ __forceinline__ double rdlane(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// test id -> cycles for REP dependent repetitions
template <int T>
__global__ void k_lat(const double *in, double *out, long long *cyc, int REP)
{
    __shared__ double sm[256];
    const int l = threadIdx.x;
    double x = in[l], y = in[64 + l];
    d4 acc = {x, y, x, y};
    d4 acc2 = {y, x, y, x};
    sm[l] = x; sm[64 + l] = y;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i0 = 0; i0 < REP; i0 += 16) {
#pragma unroll
      for (int i1 = 0; i1 < 16; i1++) {
        if (T == 0) acc = MFMA(x, y, acc);                                   // accumulator chain
        if (T == 1) { acc = MFMA(x, y, acc); acc2 = MFMA(y, x, acc2); }      // two independent chains
        if (T == 2) { acc = MFMA(acc[0], y, acc2); }                          // result -> A operand (different C)
        if (T == 3) { acc = MFMA(acc[0], y, acc2); acc = MFMA(acc[1], x, acc); }     // result -> A, then accumulate chain
        if (T == 4) x = fma(x, y, y);                                        // fp64 fma chain
        if (T == 5) x = dpp_mov<0xB1>(x) + y;                                // dpp + add chain
        if (T == 6) x = swap16(x);                                           // permlane16 swap + add chain
        if (T == 7) x = rdlane(x, 5) + y;                                    // readlane -> VALU chain
        if (T == 8) x = 1.0 / x + y;                                         // IEEE division chain
        if (T == 9) x = __builtin_amdgcn_rcp(x) + y;                         // v_rcp_f64 chain
        if (T == 10) { sm[l] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); x = sm[(l + 1) & 63] + y; }   // LDS round trip
        if (T == 11) { acc = MFMA(x, y, acc); x = acc[0] + y; }               // MFMA -> VALU -> MFMA operand
        if (T == 12) { x = fma(x, y, y); y = fma(y, x, x); acc[1] = fma(acc[1], acc[2], acc[3]); acc2[0] = fma(acc2[0], acc2[1], acc2[2]); }  // 4 fma, 2 chains of dep
        if (T == 13) x = sqrt(x * x + 1.0);
        if (T == 14) x = __shfl(x, (l + 1) & 63) + y;                        // ds_bpermute chain
        if (T == 15) { acc = MFMA(x, y, acc); x = fma(x, y, y); x = fma(x, y, y); x = fma(x, y, y); x = fma(x, y, y); }   // mfma chain + 4 dependent fma beside it
        if (T == 16) { x = fma(x, y, y); acc[1] = fma(acc[1], y, y); acc[2] = fma(acc[2], y, y); acc[3] = fma(acc[3], y, y); acc2[0] = fma(acc2[0], y, y); acc2[1] = fma(acc2[1], y, y); acc2[2] = fma(acc2[2], y, y); acc2[3] = fma(acc2[3], y, y); }   // 8 independent fma chains
        if (T == 17) x = dpp_mov<0xB1>(x);
        if (T == 18) { x = rdlane(x, 5); }
      }
    }
    long long t1 = __builtin_readcyclecounter();
    out[l] = x + y + acc[0] + acc[1] + acc[2] + acc[3] + acc2[0] + acc2[1] + acc2[2] + acc2[3];
    if (l == 0) cyc[0] = t1 - t0;
}

int main()
{
    double ha[64], hb[64], hd[256];
    srand(1);
    for (int i = 0; i < 64; i++) { ha[i] = rand() / (double)RAND_MAX; hb[i] = rand() / (double)RAND_MAX; }
    double *da, *db, *dd; long long *dc;
    hipMalloc(&da, 128 * 8); hipMalloc(&db, 64 * 8); hipMalloc(&dd, 256 * 8); hipMalloc(&dc, 8);
    hipMemcpy(da, ha, 64 * 8, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64 * 8, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, 256 * 8, hipMemcpyDeviceToHost);
    // hypothesis A: A[i=l%16][k=l/16], B[k=l/16][j=l%16]
    double ref[16][16];
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += ha[k * 16 + i] * hb[k * 16 + j]; ref[i][j] = s; }
    int okA = 0, okB = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        double v = hd[l * 4 + r];
        if (fabs(v - ref[4 * (l / 16) + r][l % 16]) < 1e-12) okA++;      // D row = 4 g + r
        if (fabs(v - ref[(l / 16) + 4 * r][l % 16]) < 1e-12) okB++;      // D row = g + 4 r
    }
    printf("layout: D[4g+r][j] matches %d/256, D[g+4r][j] matches %d/256\n", okA, okB);
    if (okA != 256 && okB != 256) {
        for (int l = 0; l < 64; l += 5) for (int r = 0; r < 4; r++) { double v = hd[l * 4 + r]; for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) if (fabs(v - ref[i][j]) < 1e-12) printf("lane %d reg %d -> (%d,%d)\n", l, r, i, j); }
    }
    double hin[128]; for (int i = 0; i < 128; i++) hin[i] = 0.5 + 0.4 * rand() / (double)RAND_MAX;
    hipMemcpy(da, hin, 128 * 8, hipMemcpyHostToDevice);
    const char *names[] = {"mfma acc chain", "2 indep mfma chains (per pair)", "mfma D->A operand", "mfma D->A then acc (per pair)", "fma chain", "dpp mov+add", "permlane16 swap+add", "readlane+add",
                           "ieee div+add", "v_rcp_f64+add", "LDS write->fence->read+add", "mfma->valu->mfma operand", "4 fma / 2 chains", "sqrt(x*x+1)", "ds_bpermute+add", "mfma chain + 4 dep fma", "8 indep fma (per 8)", "dpp mov only", "readlane only"};
#define RUN(T) do { const int REP = 2048; hipLaunchKernelGGL((k_lat<T>), dim3(1), dim3(64), 0, 0, da, dd, dc, REP); hipLaunchKernelGGL((k_lat<T>), dim3(1), dim3(64), 0, 0, da, dd, dc, REP); long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("%-36s %8.1f cycles/iter\n", names[T], (double)c / REP); } while (0)
    RUN(0); RUN(1); RUN(2); RUN(3); RUN(4); RUN(5); RUN(6); RUN(7); RUN(8); RUN(9); RUN(10); RUN(11); RUN(12); RUN(13); RUN(14); RUN(15); RUN(16); RUN(17); RUN(18);
    return 0;
}
