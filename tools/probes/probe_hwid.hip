// Where do the two wavefronts of a 128-thread workgroup land?  (HW_ID: wave_id[3:0], simd_id[5:4], cu_id[11:8], sh_id[12], se_id[15:13])
// hipcc --offload-arch=gfx950 -O2 probe_hwid.hip -o probe_hwid && ./probe_hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(128) void k(unsigned *out, long long spin)
{
    extern __shared__ double sm[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // XCC_ID
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) sm[threadIdx.x] += 1.0;
}
int main()
{
    const int B = 1024;
    unsigned *d; hipMalloc(&d, B * 4 * sizeof(unsigned));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40000);
    hipLaunchKernelGGL(k, dim3(B), dim3(128), 40000, 0, d, 2000000LL);
    hipDeviceSynchronize();
    std::vector<unsigned> h(B * 4); hipMemcpy(h.data(), d, B * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
    int same_simd = 0, adjacent = 0, par_ok = 0, slot_same = 0;
    for (int b = 0; b < B; b++) {
        unsigned a0 = h[b * 4], a1 = h[b * 4 + 2];
        int s0 = (a0 >> 4) & 3, s1 = (a1 >> 4) & 3, w0 = a0 & 15, w1 = a1 & 15;
        same_simd += s0 == s1; adjacent += (s0 ^ s1) == 1; slot_same += w0 == w1;
        par_ok += ((s0 + w0) & 1) != ((s1 + w1) & 1);
        if (b < 24) printf("wg %3d: wave0 xcc %u se %u cu %2u simd %d slot %d | wave1 se %u cu %2u simd %d slot %d\n", b, h[b * 4 + 1], (a0 >> 13) & 7, (a0 >> 8) & 15, s0, w0, (a1 >> 13) & 7, (a1 >> 8) & 15, s1, w1);
    }
    printf("same simd %d  adjacent %d  same slot %d  parity differs %d of %d\n", same_simd, adjacent, slot_same, par_ok, B);
    // sweepers per SIMD (xcc, se, cu, simd) under a rule for "which wave of the workgroup sweeps"
    for (int rule = 0; rule < 4; rule++) {
        std::vector<int> cnt(8 * 8 * 16 * 4, 0), waves(8 * 8 * 16 * 4, 0);
        int conflicts = 0;
        for (int b = 0; b < B; b++) {
            int par[2], key[2];
            for (int w = 0; w < 2; w++) {
                unsigned a = h[b * 4 + 2 * w], xcc = h[b * 4 + 2 * w + 1];
                int simd = (a >> 4) & 3, slot = a & 15, cu = (a >> 8) & 15, se = (a >> 13) & 7;
                key[w] = ((xcc * 8 + se) * 16 + cu) * 4 + simd;
                waves[key[w]]++;
                par[w] = rule == 0 ? w : rule == 1 ? ((slot + (simd >> 1)) & 1) : rule == 2 ? ((slot + simd) & 1) : (slot & 1) ^ ((simd >> 1) & 1) ^ (simd & 1);
            }
            int sw = 0;
            if (par[0] != par[1]) sw = par[0] == 0 ? 0 : 1; else conflicts++;
            cnt[key[sw]]++;
        }
        int hist[5] = {0}, used = 0, wh[5] = {0};
        for (size_t i = 0; i < cnt.size(); i++) if (waves[i]) { used++; hist[cnt[i] > 4 ? 4 : cnt[i]]++; wh[waves[i] > 4 ? 4 : waves[i]]++; }
        printf("rule %d: conflicts %d; SIMDs in use %d (waves per SIMD 1:%d 2:%d 3:%d); sweepers per SIMD 0:%d 1:%d 2:%d 3+:%d\n", rule, conflicts, used, wh[1], wh[2], wh[3], hist[0], hist[1], hist[2], hist[3] + hist[4]);
    }
    return 0;
}
