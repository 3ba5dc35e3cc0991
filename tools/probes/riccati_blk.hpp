// riccati_blk.hpp (PROBE, not part of the product: measured in round 4 and not taken, NOTES.md R4) -- the backward Riccati sweep of the interior-point QP on a PARTIALLY CONDENSED horizon (blocks of two shooting intervals):
// what the reference asks HPIPM for by name (qp_solver = "PARTIAL_CONDENSING_HPIPM", python/main.py:229) -- N = 40 stages become 20 block stages
// with the state of the odd stages eliminated, so every sequential sweep of an interior-point iteration (factor, vector, forward) has half the
// dependent stages.  The Newton system is the same (same step up to rounding: tools/probes/block2_newton_model.py).
//
// Block m = stages (a, b) = (2m, 2m + 1); block variables (x_a, u_a, u_b); x_b = Y (x_a, u_a) + rb_a with Y = [A_a B_a];
//     block dynamics  x_{a+2} = At x_a + Bt (u_a, u_b) + rt,    At = A_b A_a,  Bt = [A_b B_a | B_b],  rt = A_b rb_a + rb_b
//     block Hessian   H~_a (+) rows of stage b pulled back through Y:  sum_r d_r rho_r rho_r',  rho_r = E' v_r
// The stage cost is a weighted sum of squares of rows v_r (LINEAR_LS with a diagonal weight matrix, python/mpc.py:49-64: eight state rows, two input
// rows, two rows x_act - u) and every constraint row has the direction of one of them (boxes, rate rows u - x_act, python/mpc.py:79-99), so the
// barrier-augmented Hessian of stage b is  sum_r (w_r + gamma_r) v_r v_r'  and its pull-back is ONE matrix-core contraction over the rows:
//     tile += [rho_0 .. rho_9]' diag(d) [rho_0 .. rho_9]      3 x v_mfma_f64_16x16x4_f64   (K = 8 state rows + 2 difference rows, padded to 12)
// with the gradient of stage b riding on the affine column of the scaled operand (d_r rho_r[aff] + gt_b[r]).  Then, as in riccati_mfma.hpp,
//     W = P+ [At Bt rt] + p+          2 x v_mfma
//     G = tile + [At Bt rt]' W        2 x v_mfma
//     K = Guu^-1 [Gux kff]            1 x v_mfma, Guu^-1 (4 x 4) formed from ten v_readlane pairs by its 2 x 2 blocks
//     [P p; Mt ct] = [Gxx; At rt] - [Gxu; Bt] K     1 x v_mfma
// nine matrix instructions per TWO shooting intervals (ten in the stage-wise sweep) and ONE round of operand preparation, gain formation and
// stores instead of two.
//
// Tile layout: lane = 16 g + j; A operand A[i = j][k = g]; B operand B[k = g][j]; result register r: D[g + 4 r][j].
//     t = 0..7 state   t = 8, 9 u_a   t = 10, 11 u_b   t = 12 affine column   t = 13..15 unused (exact zeros in the records)
// Per-block record (BREC doubles, written once per solve by the condensing phase; the residual entries follow the iterate):
//     Psi tile (8 x 16):   row i = [At[i][0..7] | Bt[i][0..3] | rt[i] | 0 0 0]
//     Y tile  (12 x 16):   rows 0..7 = [A_a[i][0..7] | B_a[i][0..1] | 0 0 | rb_a[i] | 0 0 0];  rows 8, 9 = row 6 + c with -1 at t = 10 + c;  rows 10, 11 = 0
#pragma once
#include "riccati_mfma.hpp"

namespace ihm2 {

#define BREC 320            // doubles per block record
#define BREC_PSI 0
#define BREC_Y 128
#define BREC_RT 12          // column of rt / rb_a inside a tile row

// LDS arrays of the block sweep: offsets in doubles from the start of the block's dynamic LDS
struct BlkLds {
    int gam;      // (NS,NCK) barrier weights per constraint row (original stages)
    int gt;       // (NS,10)  in: modified gradient (original stages)
    int pv;       // (NS,8)   out: p at the even stages and at N
    int hv;       // (N/2,8)  out: P+ rt per block
    int Kl;       // (N/2,32) out: K per block (4 x 8)
    int Ginv;     // (N/2,16) out: Guu^-1 per block (4 x 4)
    int kff;      // (N/2,4)  out: kff per block
    int dz;       // (NS,10)  out: dz[(2m+2)*10 + i] = ct_m[i]
};

// In (LDS): gam, gt.  In (HBM/L2): brec (N/2, BREC).  Out: LDS arrays of BlkLds; Pg (NS,64): P at the even stages and N (RIC_IDX layout);
// Mg (N/2,64): Mt per block.  Hs (NS,10,10), CD (N,2,10): batch-shared, identical for all k < N; wd (12): the diagonal of cost_scale * W
// (rows: 8 states, 2 inputs, 2 differences x_act - u).  D: depth of the record ring (blocks ahead).
template <int NCK, int D>
__device__ __forceinline__ void riccati_sweep_blk2(const int N, const int lane, const double *__restrict__ brec, const double *__restrict__ Hs,
                                                   const double *__restrict__ CD, const double *__restrict__ wd, const BlkLds L,
                                                   double *__restrict__ Pg, double *__restrict__ Mg, const bool store_p)
{
    const int NB = N >> 1;
    const int g = lane >> 4, j = lane & 15;
    extern __shared__ double ric_sm[];
#define sm ric_sm
    // ---- per-lane constants ----
    // tile index -> stage-a variable (0..9), -1 otherwise
    const int col_a = (j < 10) ? j : -1;
    const int row_a[3] = {g, g + 4, (g < 2) ? 8 + g : -1};              // result registers 0, 1, 2 as rows of stage a (register 2: u_a for g = 0, 1)
    const int ub = (g >= 2) ? g - 2 : -1;                                // register 2 of the lane groups 2, 3: row u_b[ub] (tile row 10 + ub)
    const double m12 = (j == BREC_RT) ? 1.0 : 0.0;
    // C operand of the block, register r: hconst + dsel * val + g10 * cc0 + g11 * cc1; val: barrier weight on the diagonal, gradient on the affine column
    double hconst[3], cc0[3], cc1[3], dsel[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const bool ok = row_a[r] >= 0 && col_a >= 0;
        hconst[r] = ok ? Hs[row_a[r] * 10 + col_a] : 0.0;
        cc0[r] = ok ? CD[row_a[r]] * CD[col_a] : 0.0;
        cc1[r] = ok ? CD[10 + row_a[r]] * CD[10 + col_a] : 0.0;
        dsel[r] = ((ok && row_a[r] == col_a) || (row_a[r] >= 0 && j == BREC_RT)) ? 1.0 : 0.0;
    }
    if (ub >= 0) {          // u_b rows: the input rows of stage b alone (its difference rows come in through the contraction)
        hconst[2] = (j == 10 + ub) ? wd[8 + ub] : 0.0;
        dsel[2] = (j == 10 + ub || j == BREC_RT) ? 1.0 : 0.0;
    }
    // weights of the pulled-back rows of stage b: chunk 0 rows g, chunk 1 rows 4 + g (state rows), chunk 2 rows (difference 0, difference 1, -, -)
    const double wx0 = wd[g], wx1 = wd[4 + g], wx2 = (g < 2) ? wd[10 + g] : 0.0, m2 = (g < 2) ? 1.0 : 0.0;
    // LDS addresses of the last block (m = NB - 1: a = N - 2, b = N - 1), stepped down by two stages per block
    const int a0 = N - 2, b0 = N - 1;
    int va01 = ((j == BREC_RT) ? L.gt + a0 * 10 : L.gam + a0 * NCK) + g;                                     // rows g, g + 4 of stage a
    int va2 = (ub < 0) ? ((j == BREC_RT) ? L.gt + a0 * 10 : L.gam + a0 * NCK) + 8 + ((g < 2) ? g : 0)
                       : ((j == BREC_RT) ? L.gt + b0 * 10 : L.gam + b0 * NCK) + 8 + ub;                      // row u_a[g] / u_b[ub]
    const int st01 = (j == BREC_RT) ? 20 : 2 * NCK, st2 = st01;
    int a_g10 = L.gam + a0 * NCK + 10;                  // rate-row weights of stage a
    int a_wb = L.gam + b0 * NCK;                        // barrier weights of stage b: rows g, 4 + g, 10 + g
    int a_gb = L.gt + b0 * 10;                          // gradient of stage b (state part): rows g, 4 + g
    // M-row operand of the last product: Bt[j - 8][g] for the lanes j >= 8
    const unsigned offBt = (j >= 8) ? (unsigned)((j - 8) * 16 + 8 + g) : 0u;
    // A operand of the gain product: Ginv[j][g] for j < 4 -- selected from the ten entries of the symmetric inverse by lane masks (scalar registers)
    const int qa = (j < 4) ? ((j <= g) ? j * 4 - j * (j - 1) / 2 + (g - j) : g * 4 - g * (g - 1) / 2 + (j - g)) : -1;      // index of (min, max) in the packed upper triangle

    // ---- terminal stage: P_N = H~_N (state block), p_N = gradient ----
    d4_t Pd;
    {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int row = g + 4 * r;
            double v = 0.0;
            if (j < 8) {
                v = Hs[(N * 10 + row) * 10 + j];
                if (row == j) v += sm[L.gam + N * NCK + j];
            } else if (j == BREC_RT) v = sm[L.gt + N * 10 + row];
            Pd[r] = v;
            if (j < 8) Pg[(size_t)N * 64 + RIC_IDX(row, j)] = v;
            if (j == BREC_RT) sm[L.pv + N * 8 + row] = v;
        }
        Pd[2] = Pd[3] = 0.0;
    }

    // ---- record ring: six loads per lane and block, D blocks ahead ----
    double rp0[D], rp1[D], ry0[D], ry1[D], ry2[D], rBt[D];
    auto ring_load = [&](const int d, const int m) {
        const double *rec = brec + (size_t)max(m, 0) * BREC;
        rp0[d] = rec[BREC_PSI + lane]; rp1[d] = rec[BREC_PSI + 64 + lane];
        ry0[d] = rec[BREC_Y + lane]; ry1[d] = rec[BREC_Y + 64 + lane]; ry2[d] = rec[BREC_Y + 128 + lane];
        rBt[d] = rec[BREC_PSI + offBt];
    };
#pragma unroll
    for (int d = 0; d < D; d++) ring_load(d, NB - 1 - d);

    // LDS operands of a block, fetched one block ahead
    double val0, val1, val2, g10, g11, w0, w1, w2, gb0, gb1;
    auto prepare_load = [&]() {
        val0 = sm[va01]; val1 = sm[va01 + 4]; val2 = sm[va2];
        g10 = sm[a_g10]; g11 = sm[a_g10 + 1];
        w0 = sm[a_wb + g]; w1 = sm[a_wb + 4 + g]; w2 = sm[a_wb + 10 + (g & 1)];
        gb0 = sm[a_gb + g]; gb1 = sm[a_gb + 4 + g];
        va01 -= st01; va2 -= st2; a_g10 -= 2 * NCK; a_wb -= 2 * NCK; a_gb -= 20;
    };
    prepare_load();
    RIC_STAMP_DECL
    for (int s0 = 0; s0 < NB; s0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int s = s0 + d;
            const int m = NB - 1 - s, a = 2 * m;
            if (s < NB) {
                RIC_STAMP(0);
                const double p0 = rp0[d], p1 = rp1[d], y0 = ry0[d], y1 = ry1[d], y2 = ry2[d], Btl = rBt[d];
                // ---- block Hessian and gradient: stage a directly, stage b through its rows ----
                d4_t T;
                T[0] = fma(g11, cc1[0], fma(g10, cc0[0], fma(dsel[0], val0, hconst[0])));
                T[1] = fma(g11, cc1[1], fma(g10, cc0[1], fma(dsel[1], val1, hconst[1])));
                T[2] = fma(g11, cc1[2], fma(g10, cc0[2], fma(dsel[2], val2, hconst[2])));
                T[3] = 0.0;
                const double s0b = fma(wx0 + w0, y0, m12 * gb0), s1b = fma(wx1 + w1, y1, m12 * gb1), s2b = (m2 * (wx2 + w2)) * y2;
                T = IHM2_MFMA_F64(y0, s0b, T);
                T = IHM2_MFMA_F64(y1, s1b, T);
                T = IHM2_MFMA_F64(y2, s2b, T);
                RIC_STAMP(1);
                // ---- W = P+ [At Bt rt] + p+ on the affine column ----
                const double q0 = m12 * Pd[0], q1 = m12 * Pd[1];
                d4_t W = {q0, q1, 0.0, 0.0};
                W = IHM2_MFMA_F64(Pd[0], p0, W);
                W = IHM2_MFMA_F64(Pd[1], p1, W);
                // ---- G = tile + [At Bt rt]' W ----
                d4_t G = T;
                G = IHM2_MFMA_F64(p0, W[0], G);
                G = IHM2_MFMA_F64(p1, W[1], G);
                RIC_STAMP(2);
                // ---- Guu^-1 (4 x 4, symmetric positive definite) by its 2 x 2 blocks [E F; F' H]; every lane forms all of it ----
                const double e0 = readlane_f64(G[2], 8), e1 = readlane_f64(G[2], 9), e2 = readlane_f64(G[2], 25);
                const double f0 = readlane_f64(G[2], 10), f1 = readlane_f64(G[2], 11), f2 = readlane_f64(G[2], 26), f3 = readlane_f64(G[2], 27);
                const double h0 = readlane_f64(G[2], 42), h1 = readlane_f64(G[2], 43), h2 = readlane_f64(G[2], 59);
                const double dE = e0 * e2 - e1 * e1;
                double iE = __builtin_amdgcn_rcp(dE);
                iE = fma(fma(-dE, iE, 1.0), iE, iE);
                iE = fma(fma(-dE, iE, 1.0), iE, iE);
                const double E0 = e2 * iE, E1 = -e1 * iE, E2 = e0 * iE;                             // E^-1
                const double x0 = E0 * f0 + E1 * f2, x1 = E0 * f1 + E1 * f3, x2 = E1 * f0 + E2 * f2, x3 = E1 * f1 + E2 * f3;        // X = E^-1 F
                const double s0 = h0 - (f0 * x0 + f2 * x2), s1 = h1 - (f0 * x1 + f2 * x3), s2 = h2 - (f1 * x1 + f3 * x3);          // S = H - F' X
                const double dS = s0 * s2 - s1 * s1;
                double iS = __builtin_amdgcn_rcp(dS);
                iS = fma(fma(-dS, iS, 1.0), iS, iS);
                iS = fma(fma(-dS, iS, 1.0), iS, iS);
                const double S0 = s2 * iS, S1 = -s1 * iS, S2 = s0 * iS;                             // S^-1: lower right block of the inverse
                const double r0 = -(x0 * S0 + x1 * S1), r1 = -(x0 * S1 + x1 * S2), r2 = -(x2 * S0 + x3 * S1), r3 = -(x2 * S1 + x3 * S2);    // upper right: -X S^-1
                const double t0 = E0 - (r0 * x0 + r1 * x1), t1 = E1 - (r0 * x2 + r1 * x3), t2 = E2 - (r2 * x2 + r3 * x3);                  // upper left: E^-1 + X S^-1 X'
                // packed upper triangle (row-major): (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
                const double tri[10] = {t0, t1, r0, r1, t2, r2, r3, S0, S1, S2};
                double Ai = 0.0;
#pragma unroll
                for (int q = 0; q < 10; q++) Ai = (qa == q) ? tri[q] : Ai;
                // ---- K = Guu^-1 [G(u,:)]: rows u_a0, u_a1, u_b0, u_b1 sit in register 2 of the four lane groups ----
                d4_t Kt = {0.0, 0.0, 0.0, 0.0};
                Kt = IHM2_MFMA_F64(Ai, G[2], Kt);
                const double Kf = Kt[0];
                RIC_STAMP(3);
                // ---- [P p; Mt ct] = [Gxx; At rt] - [Gxu; Bt] K ----
                const double Sa = (j < 8) ? G[2] : Btl;
                d4_t S = {G[0], G[1], p0, p1};
                S = IHM2_MFMA_F64(Sa, -Kf, S);
                RIC_STAMP(4);
                ring_load(d, m - D);
                if (m > 0) prepare_load();
                RIC_STAMP(5);
                if (j < 8) {
                    double2 mm;
                    mm.x = S[2]; mm.y = S[3];
                    *(double2 *)(Mg + (size_t)m * 64 + RIC_IDX(g, j)) = mm;
                    sm[L.Kl + m * 32 + g * 8 + j] = Kf;
                    if (j < 4) sm[L.Ginv + m * 16 + j * 4 + g] = Ai;
                }
                if (j == BREC_RT) {
                    sm[L.hv + m * 8 + g] = W[0] - q0; sm[L.hv + m * 8 + 4 + g] = W[1] - q1;        // P+ rt
                    if (store_p) { sm[L.pv + a * 8 + g] = S[0]; sm[L.pv + a * 8 + 4 + g] = S[1]; }
                    sm[L.dz + (a + 2) * 10 + g] = S[2]; sm[L.dz + (a + 2) * 10 + 4 + g] = S[3];
                    sm[L.kff + m * 4 + g] = Kf;
                }
                RIC_WSYNC();
                Pd[0] = S[0]; Pd[1] = S[1];
                if (j < 8) {
                    double2 pp;
                    pp.x = Pd[0]; pp.y = Pd[1];
                    *(double2 *)(Pg + (size_t)a * 64 + RIC_IDX(g, j)) = pp;
                }
                RIC_STAMP(6);
            }
        }
    }
    RIC_STAMP_OUT;
#undef sm
}

}  // namespace ihm2
