// riccati_mfma.hpp -- the backward Riccati sweep of the interior-point QP (HOT LOOP 3: the KKT factor of what HPIPM solves inside
// AcadosOcpSolver.solve(), python/main.py:228-233,325) as a chain of fp64 matrix-core products that never leaves the registers.
//
// One wavefront per instance.  Per stage k = N-1 .. 0, with the records [A|B|b|rb]_k streamed from HBM/L2:
//     W  = P_{k+1} [A B rb] + [0 0 p_{k+1}]           2 x v_mfma_f64_16x16x4_f64   (contraction over the 8 states)
//     G  = [H~_k g_k] + [A B rb]' W                    2 x v_mfma                   (10x10 Hessian block + gradient column)
//     [P_k p_k | M_k c_k] = [Gxx | A rb] - [Gxu | B] K  1 x v_mfma, K = Guu^-1 [Gux kff]
// so the factorisation, the PREDICTOR's vector recursion p_k, its feed-forward kff_k, the affine part c_k = rb_k - B kff_k of the
// forward recursion and the closed-loop matrix M_k = A - B K come out of the same five matrix instructions, with no LDS hand-off
// and no lane exchange between them (the old VALU/LDS sweep spent 2480 cycles per stage on three hand-offs).
//
// What the sweep is priced in (measured, tools/probes/probe_mfma_f64.hip, probe_mfma_shadow.hip): an fp64 MFMA holds the SIMD for its
// 64 cycles -- NOTHING of the same wave issues in its shadow (vector, LDS and memory instructions all add their full issue time:
// ~5 cycles per vector instruction, ~11 per LDS write, ~14 per global store) -- and fp64 matrix rate = fp64 vector rate on gfx950,
// so the matrix core buys no flops here; it buys the DATA MOVEMENT: operands and results of consecutive products are in the same
// registers.  Everything else in the stage is therefore kept to per-lane constants, two-instruction selects and ring loads:
// the dynamics residual rb_k sits in the record itself (slot 88..95, written by dyn_residual for all stages in parallel).
//
// Tile layout (measured): lane = 16 g + j.  A operand: A[i = j][k = g]; B operand: B[k = g][j]; result register r: D[i = g + 4 r][j].
// The result rows 0..7 of a product are laid out as the two K-chunks of the next product's B operand (chunk c = register c), and
// -- P being symmetric -- as the two chunks of its A operand.  Tile index t (row or column):
//     t = 0..7  state i       t = 8, 9  input u0, u1       t = 12, 13  u1, u0 AGAIN (lane group 0 then holds the rows u0, u1 and
//     t = 10    affine column: rb -> P rb + p -> gradient -> p_k, kff_k, c_k         group 1 the rows u1, u0 of G: K without a swap)
//     t = 11, 14, 15  unused: finite junk that never reaches a used entry
#pragma once
// diagnostic cycle stamps (tools/probes/riccati_proto.hip -DRIC_STAMPS): time between consecutive marks, accumulated per section
#ifdef RIC_STAMPS
__device__ long long ric_dbg[8];
#define RIC_STAMP_DECL long long ric_T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ric_tp = __builtin_readcyclecounter(); int ric_cur = 7;
#define RIC_STAMP(i) do { const long long t_now = __builtin_readcyclecounter(); ric_T[ric_cur] += t_now - ric_tp; ric_tp = t_now; ric_cur = (i); } while (0)
#define RIC_STAMP_OUT do { if (threadIdx.x == 0 && blockIdx.x == 0) for (int q = 0; q < 8; q++) ric_dbg[q] = ric_T[q]; } while (0)
#else
#define RIC_STAMP_DECL
#define RIC_STAMP(i)
#define RIC_STAMP_OUT
#endif

#include <hip/hip_runtime.h>

namespace ihm2 {

typedef double d4_t __attribute__((ext_vector_type(4)));
#define IHM2_MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// one wavefront per block: an LDS hand-off needs no s_barrier, only a compiler fence (the LDS serves a wave's instructions in order)
#define RIC_WSYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

// Lane exchanges at VALU speed (no LDS crossbar): DPP moves inside a 16-lane row.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every lane of these permutations has an in-range source, so the "old" operand is never used: the mov_dpp form leaves it undefined,
    // which saves the two register copies per move that a tied old = source costs (v_mov_b32_dpp writes a fresh register)
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// P_k and M_k in HBM/L2: entry (row, col) of an 8x8 block at RIC_IDX -- the two result registers of a lane are adjacent, so a
// stage's block is written by one 16-byte store per lane (lanes j < 8)
#define RIC_IDX(row, col) (((((row) & 3) * 8 + (col)) << 1) + ((row) >> 2))
#define RIC_REC_RB 88       // slot of rb_k in a linearisation record

// LDS arrays of the instance: offsets in doubles from the start of the block's dynamic LDS (the sweep addresses them through its
// own extern __shared__ declaration, i.e. as LDS for certain -- a generic pointer handed in through a struct made the compiler
// emit flat address arithmetic in the register-starved instantiations)
struct RicLds {
    int gam;      // (NS,NCK) barrier weights per constraint row
    int hc;       // (NS,2)   track-row slopes (PATH)
    int ha;       // (NS,4)   non-zeros of the lateral-acceleration row in (v_x, v_y, T, delta) (ALAT; zeros where the row is absent)
    int gt;       // (NS,10)  in: modified gradient of the predictor
    int pv;       // (NS,8)   out: p_N, and p_k (k < N) if store_p
    int hv;       // (N,8)    out: P_{k+1} rb_k
    int Kl;       // (N,16)   out: K_k
    int Ginv;     // (N,8)    out: Guu^-1 as (Gi0, Gi1, Gi2, Gi1, 0 ..)
    int kff;      // (N,4)    out: kff_k (2 used)
    int dz;       // (NS,10)  out: dz[(k+1)*10 + i] = c_k[i], i < 8
    int tile;     // (8,17)   scratch: the transpose of P_k (rows padded to 17 against bank conflicts)
};

// rb_k = A_k z_k + B_k u_k + b_k - z_{k+1} (-> LDS and the record's slot) and gt_k += [A B]_k' pi_{k+1}, all stages in parallel:
// one dot product per lane, no reductions.  The record entries of a BATCH of dot products are loaded before any of them is used
// (the wave is alone on its SIMD: a load that is consumed at once costs its full L2 latency, and the compiler cannot batch the
// loads itself across the stores in between).  What bounds the phase is the number of cache lines its scattered loads touch (a
// lane's row / column of [A B] lies in other lines than its neighbour's); staging the records through LDS with coalesced loads
// was tried and lost to LDS bank conflicts (rows 64 bytes apart: 16-way), measured 29 k cycles against 10 k.
// NT: threads that share the pass (64 for a wavefront per instance, 64 NW for the block kernel); lane = thread index among them
template <int NB, int NT = 64>
__device__ __forceinline__ void dyn_residual(const int N, const int lane, double *linb, const double *z, const double *pi, double *gt, double *rb,
                                             const int lin_rec)
{
    const int n1 = N * 8;
    for (int base = 0; base < n1; base += NT * NB) {
        double2 Ar[NB][4], Br[NB];
        double br[NB];
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const int e = min(base + NT * q + lane, n1 - 1), k = e >> 3, o = e & 7;
            const double *rec = linb + (size_t)k * lin_rec;
            const double2 *row = reinterpret_cast<const double2 *>(rec + o * 8);
#pragma unroll
            for (int l = 0; l < 4; l++) Ar[q][l] = row[l];
            Br[q] = *reinterpret_cast<const double2 *>(rec + 64 + o * 2);
            br[q] = rec[80 + o];
        }
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const int e = base + NT * q + lane, ec = min(e, n1 - 1), k = ec >> 3, o = ec & 7;
            double acc = br[q] - z[(k + 1) * 10 + o];
#pragma unroll
            for (int l = 0; l < 4; l++) { acc = fma(Ar[q][l].x, z[k * 10 + 2 * l], acc); acc = fma(Ar[q][l].y, z[k * 10 + 2 * l + 1], acc); }
            acc = fma(Br[q].x, z[k * 10 + 8], acc);
            acc = fma(Br[q].y, z[k * 10 + 9], acc);
            if (e < n1) {
                rb[e] = acc;
                linb[(size_t)k * lin_rec + RIC_REC_RB + o] = acc;
            }
        }
    }
    const int n2 = N * 10;
    for (int base = 0; base < n2; base += NT * NB) {
        double cr[NB][8];
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const int e = min(base + NT * q + lane, n2 - 1), k = e / 10, jz = e - k * 10;
            const double *rec = linb + (size_t)k * lin_rec;
            const double *col = (jz < 8) ? rec + jz : rec + 64 + (jz - 8);
            const int cs = (jz < 8) ? 8 : 2;
#pragma unroll
            for (int l = 0; l < 8; l++) cr[q][l] = col[l * cs];
        }
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const int e = base + NT * q + lane, ec = min(e, n2 - 1), k = ec / 10;
            double acc = gt[ec];
#pragma unroll
            for (int l = 0; l < 8; l++) acc = fma(cr[q][l], pi[(k + 1) * 8 + l], acc);
            if (e < n2) gt[e] = acc;
        }
    }
}

// Backward sweep.  In (LDS): gam, hc, gt = the predictor's gradient incl. [A B]'pi; in the records: A, B and rb (slot 88).
// Out: LDS arrays of RicLds; HBM: Pg (NS,64), Mg (N,64) in the RIC_IDX layout.  store_p (wave-uniform): also keep p_k, k < N.
// Hs (NS,10,10), CD (N,2,10): batch-shared; UNI: the same for all k < N.  D: depth of the record prefetch ring.
// ALAT: a fifteenth row per stage, gam[k][14] a a' with a = the four non-zeros in L.ha (stages 1..N-1).
template <int NCK, bool PATH, bool UNI, int D, bool ALAT = false>
__device__ __forceinline__ void riccati_sweep_mfma(const int N, const int lane, const double *__restrict__ linb, const double *__restrict__ Hs,
                                                   const double *__restrict__ CD, const RicLds L, double *__restrict__ Pg, double *__restrict__ Mg,
                                                   const int lin_rec, const bool store_p, const bool symmetrize)
{
    const int g = lane >> 4, j = lane & 15;
    const int col = (j < 10) ? j : (j == 12) ? 9 : (j == 13) ? 8 : -1;
    const int row[4] = {g, g + 4, (g < 2) ? 8 + g : -1, (g == 0) ? 9 : (g == 1) ? 8 : -1};
    const double m10 = (j == 10) ? 1.0 : 0.0;
    // the block's dynamic LDS, indexed directly: every access below is a ds_ instruction by construction (a pointer variable, even an
    // LDS-typed one, made the register-starved instantiations go through generic-pointer casts the compiler then mis-folded)
    extern __shared__ double ric_sm[];
#define sm ric_sm

    // ---- per-lane constants: record offsets, Hessian entries, LDS addresses ----
    unsigned off[2], offB;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int rr = 4 * c + g;
        off[c] = (j < 8) ? rr * 8 + j : (col >= 0) ? 64 + rr * 2 + (col - 8) : RIC_REC_RB + rr;      // unused columns: finite junk
    }
    offB = (j >= 8) ? 64 + (j - 8) * 2 + (g & 1) : 0;
    // C operand of the G product, result register r: hconst + dsel * val + gam10 * cc0 + gam11 * cc1 (+ track-row terms)
    // val: gam[k][row] on the diagonal; the gradient column t = 10 reads gt[k][row] instead
    double hconst[4], cc0[4], cc1[4], dsel[4];
    int va[4], vstride[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const bool ok = row[r] >= 0 && col >= 0, gr = row[r] >= 0 && j == 10;
        hconst[r] = (UNI && ok) ? Hs[row[r] * 10 + col] : 0.0;
        cc0[r] = (UNI && ok) ? CD[row[r]] * CD[col] : 0.0;
        cc1[r] = (UNI && ok) ? CD[10 + row[r]] * CD[10 + col] : 0.0;
        dsel[r] = ((ok && row[r] == col) || gr) ? 1.0 : 0.0;
        // (rows 8 + g of the lane groups 2, 3 do not exist: their dsel is 0 and the address g + 8 stays inside the arrays)
        vstride[r] = (j == 10) ? 10 : NCK;
        va[r] = ((j == 10) ? L.gt : L.gam) + ((r == 3) ? ((row[3] >= 0) ? row[3] : 0) : g) + (N - 1) * vstride[r];
    }
    int a_g10 = L.gam + (N - 1) * NCK + 10;
    // track rows (PATH): entries (n,n), (n,psi), (psi,psi) of H~ -- result register 0 of the lane groups 1, 2
    const double p11 = (PATH && row[0] == 1 && col == 1) ? 1.0 : 0.0, p22 = (PATH && row[0] == 2 && col == 2) ? 1.0 : 0.0;
    const double p12 = (PATH && ((row[0] == 1 && col == 2) || (row[0] == 2 && col == 1))) ? 1.0 : 0.0;
    // lateral-acceleration row (ALAT): gam14 a_i a_l on the entries (i, l) of {v_x, v_y, T, delta}^2 -- state rows sit in the result registers 0, 1
    auto a_slot = [](int c) -> int { return (c == 3) ? 0 : (c == 4) ? 1 : (c == 6) ? 2 : (c == 7) ? 3 : -1; };
    const int as_c = ALAT ? a_slot(col) : -1, as_r0 = ALAT ? a_slot(row[0]) : -1, as_r1 = ALAT ? a_slot(row[1]) : -1;
    const double am0 = (as_c >= 0 && as_r0 >= 0) ? 1.0 : 0.0, am1 = (as_c >= 0 && as_r1 >= 0) ? 1.0 : 0.0;
    const int ao_c = L.ha + max(as_c, 0), ao_r0 = L.ha + max(as_r0, 0), ao_r1 = L.ha + max(as_r1, 0);
    // K = Guu^-1 [G(u0,:); G(u1,:)]: lane group 0 holds (u0, u1) in the result registers (2, 3), group 1 holds (u1, u0)
    const double selA0 = (g == 0) ? 1.0 : 0.0, selA2 = (g == 1) ? 1.0 : 0.0, selB = (g < 2) ? -1.0 : 0.0;

    // transpose tile: lane (g, j) writes rows g, g + 4 at column j and reads (row j & 7, columns g, g + 4); the affine column keeps its value
    const int a_tw = L.tile + g * 17 + j, a_tr = L.tile + (j & 7) * 17 + g;
    const double wS = (j < 8) ? 0.5 : 1.0, wT = (j < 8) ? 0.5 : 0.0;

    // ---- terminal stage: P_N = H~_N (state block), p_N = gradient ----
    d4_t Pd;
    {
        const double g12 = PATH ? sm[L.gam + N * NCK + 12] : 0.0, g13 = PATH ? sm[L.gam + N * NCK + 13] : 0.0;
        const double a0 = PATH ? sm[L.hc + N * 2] : 0.0, a1 = PATH ? sm[L.hc + N * 2 + 1] : 0.0;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            double v = 0.0;
            if (j < 8) {
                v = Hs[(N * 10 + row[r]) * 10 + j];
                if (row[r] == j) v += sm[L.gam + N * NCK + j];
            } else if (j == 10) v = sm[L.gt + N * 10 + row[r]];
            if (PATH && r == 0) v += p11 * (g12 + g13) + p12 * (g12 * a0 - g13 * a1) + p22 * (g12 * a0 * a0 + g13 * a1 * a1);
            Pd[r] = v;
            if (j < 8) Pg[(size_t)N * 64 + RIC_IDX(row[r], j)] = v;
            if (j == 10) sm[L.pv + N * 8 + row[r]] = v;
        }
        Pd[2] = Pd[3] = 0.0;
    }

    // ---- record ring: three loads per lane and stage, D stages ahead (indices clamped, loads unconditional) ----
    double r0[D], r1[D], rB[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const double *rec = linb + (size_t)max(N - 1 - d, 0) * lin_rec;
        r0[d] = rec[off[0]]; r1[d] = rec[off[1]]; rB[d] = rec[offB];
    }
    // C operand of a stage's G product, prepared one stage ahead in two halves: the LDS reads are issued right after the stage's last
    // matrix instruction (before the stage's own LDS stores, whose addresses the compiler cannot tell apart), the arithmetic follows
    // once the transposed P has come back -- the reads' latency and the tile's round trip are covered by the stage's stores
    d4_t Hc;
    double val[4], g10, g11, pg12 = 0.0, pg13 = 0.0, pa0 = 0.0, pa1 = 0.0, pg14 = 0.0, pac = 0.0, par0 = 0.0, par1 = 0.0;
    auto prepare_load = [&](const int k) {
        val[0] = sm[va[0]]; val[1] = sm[va[0] + 4]; val[2] = sm[va[0] + 8]; val[3] = sm[va[3]];      // rows g, g + 4, 8 + g; the mirrored input row
        g10 = sm[a_g10]; g11 = sm[a_g10 + 1];
        if (PATH) { pg12 = sm[a_g10 + 2]; pg13 = sm[a_g10 + 3]; pa0 = sm[L.hc + k * 2]; pa1 = sm[L.hc + k * 2 + 1]; }
        if (ALAT) { pg14 = sm[a_g10 + 4]; pac = sm[ao_c + k * 4]; par0 = sm[ao_r0 + k * 4]; par1 = sm[ao_r1 + k * 4]; }
        a_g10 -= NCK;
        va[0] -= vstride[0]; va[3] -= vstride[3];
    };
    auto prepare_compute = [&](const int k) {
        double hk[4], c0k[4], c1k[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { hk[r] = hconst[r]; c0k[r] = cc0[r]; c1k[r] = cc1[r]; }
        if (!UNI) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool ok = row[r] >= 0 && col >= 0;
                hk[r] = ok ? Hs[(k * 10 + row[r]) * 10 + col] : 0.0;
                c0k[r] = ok ? CD[(k * 2 + 0) * 10 + row[r]] * CD[(k * 2 + 0) * 10 + col] : 0.0;
                c1k[r] = ok ? CD[(k * 2 + 1) * 10 + row[r]] * CD[(k * 2 + 1) * 10 + col] : 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) Hc[r] = fma(g11, c1k[r], fma(g10, c0k[r], fma(dsel[r], val[r], hk[r])));
        if (PATH) Hc[0] += p11 * (pg12 + pg13) + p12 * (pg12 * pa0 - pg13 * pa1) + p22 * (pg12 * pa0 * pa0 + pg13 * pa1 * pa1);
        if (ALAT) { Hc[0] += am0 * pg14 * par0 * pac; Hc[1] += am1 * pg14 * par1 * pac; }
    };
    prepare_load(N - 1);
    prepare_compute(N - 1);
    RIC_STAMP_DECL
    for (int s0 = 0; s0 < N; s0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int s = s0 + d;
            const int k = N - 1 - s;
            if (s < N) {
                RIC_STAMP(0);
                const double B0 = r0[d], B1 = r1[d], Bmk = rB[d];
                const d4_t Hk = Hc;
                // ---- W = P_{k+1} [A B rb] + p_{k+1} on the affine column ----
                const double q0 = m10 * Pd[0], q1 = m10 * Pd[1];
                d4_t W = {q0, q1, 0.0, 0.0};
                W = IHM2_MFMA_F64(Pd[0], B0, W);
                W = IHM2_MFMA_F64(Pd[1], B1, W);
                RIC_STAMP(1);
                // ---- G = [H~ g] + [A B rb]' W ----
                d4_t G = Hk;
                G = IHM2_MFMA_F64(B0, W[0], G);
                G = IHM2_MFMA_F64(B1, W[1], G);
                RIC_STAMP(2);
                // ---- K = Guu^-1 [G(u0,:); G(u1,:)]; the adjugate part runs beside the reciprocal of the determinant ----
                const double g00 = readlane_f64(G[2], 8), g01 = readlane_f64(G[2], 9), g11u = readlane_f64(G[2], 25);
                const double det = g00 * g11u - g01 * g01;
                double idet = __builtin_amdgcn_rcp(det);       // hardware reciprocal + two Newton steps (det of an SPD block)
                const double cAa = selA0 * g11u + selA2 * g00, cBa = selB * g01;
                const double ta = cAa * G[2] + cBa * G[3];
                idet = fma(fma(-det, idet, 1.0), idet, idet);
                idet = fma(fma(-det, idet, 1.0), idet, idet);
                const double Kf = ta * idet;
                RIC_STAMP(3);
                // ---- [P_k p_k; M_k c_k] = [Gxx; A rb] - [Gxu; B] K ----
                const double Sa = (j < 8) ? G[2] : Bmk;
                d4_t S = {G[0], G[1], B0, B1};
                S = IHM2_MFMA_F64(Sa, -Kf, S);
                RIC_STAMP(4);
                // ---- P_k := (P_k + P_k') / 2 through an LDS tile: the recursion is only as good as P's symmetry (with the open-loop
                // unstable dynamic model the antisymmetric rounding noise decided whether ill-conditioned QPs converged; the
                // reference implementation evaluates symmetric pairs identically).  The tile's round trip is covered by the work
                // that does not depend on it: the ring refill, the next stage's LDS operands, this stage's stores. ----
                if (symmetrize) { sm[a_tw] = S[0]; sm[a_tw + 4 * 17] = S[1]; }
                {
                    const double *rec = linb + (size_t)max(k - D, 0) * lin_rec;
                    r0[d] = rec[off[0]]; r1[d] = rec[off[1]]; rB[d] = rec[offB];
                }
                if (k > 0) prepare_load(k - 1);
                RIC_STAMP(5);
#ifndef RIC_SKIP_A
                if (j < 8) {
                    double2 mm;
                    mm.x = S[2]; mm.y = S[3];
                    *(double2 *)(Mg + (size_t)k * 64 + RIC_IDX(g, j)) = mm;
                    if (g < 2) sm[L.Kl + k * 16 + g * 8 + j] = Kf;
                }
#endif
#ifndef RIC_SKIP_B
                if (j == 10) {
                    sm[L.hv + k * 8 + g] = W[0] - q0; sm[L.hv + k * 8 + 4 + g] = W[1] - q1;        // P_{k+1} rb_k
                    if (store_p) { sm[L.pv + k * 8 + g] = S[0]; sm[L.pv + k * 8 + 4 + g] = S[1]; }
                    sm[L.dz + (k + 1) * 10 + g] = S[2]; sm[L.dz + (k + 1) * 10 + 4 + g] = S[3];
                    sm[L.Ginv + k * 8 + 2 * g] = cAa * idet; sm[L.Ginv + k * 8 + 2 * g + 1] = cBa * idet;
                    sm[L.kff + k * 4 + g] = Kf;
                }
#endif
                RIC_WSYNC();
                if (symmetrize) {
                    const double T0 = sm[a_tr], T1 = sm[a_tr + 4];
                    Pd[0] = fma(wT, T0, wS * S[0]);
                    Pd[1] = fma(wT, T1, wS * S[1]);
                } else { Pd[0] = S[0]; Pd[1] = S[1]; }
                if (k > 0) prepare_compute(k - 1);
#ifndef RIC_SKIP_A
                if (j < 8) {
                    double2 pp;
                    pp.x = Pd[0]; pp.y = Pd[1];
                    *(double2 *)(Pg + (size_t)k * 64 + RIC_IDX(g, j)) = pp;
                }
#endif
                RIC_STAMP(6);
            }
        }
    }
    RIC_STAMP_OUT;
#undef sm
}

}  // namespace ihm2
