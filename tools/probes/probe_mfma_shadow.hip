// What issues in the shadow of a v_mfma_f64_16x16x4_f64 (64 cycles on gfx950) from the SAME wave?  One wave per SIMD.
// Each test: loop of { MFMA (accumulator chain); K independent instructions of one kind }.  64 cycles/iter = fully hidden.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

template <int T, int K>
__global__ void k_shadow(const double *in, double *out, long long *cyc, int REP)
{
    __shared__ double sm[1024];
    const int l = threadIdx.x;
    double x = in[l], y = in[64 + l];
    d4 acc = {x, y, x, y};
    int i0 = l, i1 = l + 1, i2 = l + 2, i3 = l + 3;
    float f0 = (float)x, f1 = (float)y, f2 = f0 + 1, f3 = f1 + 1;
    double d0 = x, d1 = y, d2 = x + 1, d3 = y + 1;
    int s0 = REP, s1 = REP + 1;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < REP; it += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc = MFMA(x, y, acc);
#pragma unroll
            for (int q = 0; q < K; q += 4) {
                if (T == 1) asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(l));
                if (T == 2) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
                if (T == 3) asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
                if (T == 4) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1" : "+s"(s0), "+s"(s1));
                if (T == 5) asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1536" : : "v"(l * 8), "v"(d0) : "memory");
                if (T == 6) asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(l) : "vcc");
                if (T == 7) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
                if (T == 8) asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:512\n s_waitcnt lgkmcnt(0)" : "=v"(d2), "=v"(d3) : "v"(l * 8) : "memory");
                if (T == 9) asm volatile("global_store_dwordx2 %0, %1, %2\n global_store_dwordx2 %0, %1, %2 offset:512\n global_store_dwordx2 %0, %1, %2 offset:1024\n global_store_dwordx2 %0, %1, %2 offset:1536" : : "v"(l * 8), "v"(d0), "s"(out) : "memory");
                if (T == 10) asm volatile("v_mul_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
                if (T == 11) asm volatile("v_readlane_b32 %0, %2, 3\n v_readlane_b32 %1, %2, 5\n v_readlane_b32 %0, %2, 7\n v_readlane_b32 %1, %2, 9" : "=s"(s0), "=s"(s1) : "v"(i0));
                if (T == 12) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
                if (T == 13) asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d0) : "v"(d1));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[l + 4096] = x + y + acc[0] + acc[1] + acc[2] + acc[3] + i0 + i1 + i2 + i3 + f0 + f1 + f2 + f3 + d0 + d1 + d2 + d3 + s0 + s1 + sm[l];
    if (l == 0) cyc[0] = t1 - t0;
}

int main()
{
    double hin[128]; for (int i = 0; i < 128; i++) hin[i] = 0.5 + 0.001 * i;
    double *da, *dd; long long *dc;
    hipMalloc(&da, 128 * 8); hipMalloc(&dd, 8192 * 8); hipMalloc(&dc, 8);
    hipMemcpy(da, hin, 128 * 8, hipMemcpyHostToDevice);
    const char *names[] = {"mfma only", "v_add_u32", "v_fma_f32", "v_fma_f64", "s_add_u32", "ds_write_b64", "v_cndmask_b32", "v_mov_b32_dpp", "2 ds_read_b64 + wait (per 4)", "global_store_dwordx2",
                           "v_mul/add_f64", "v_readlane_b32", "s_nop 0", "v_lshl_add_u64"};
#define RUN(T, K) do { const int REP = 2048; for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_shadow<T, K>), dim3(1), dim3(64), 0, 0, da, dd, dc, REP); long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("%-32s K=%2d  %7.1f cycles/iter\n", names[T], K, (double)c / REP); } while (0)
    RUN(0, 0);
#define ALLK(T) RUN(T, 4); RUN(T, 8); RUN(T, 12); RUN(T, 16); RUN(T, 24)
    ALLK(1); ALLK(2); ALLK(3); ALLK(4); ALLK(5); ALLK(6); ALLK(7); ALLK(8); ALLK(9); ALLK(10); ALLK(11); ALLK(12); ALLK(13);
    return 0;
}
