"""NumPy model of the partially condensed Newton solve (block size 2) of the interior-point iteration, checked against the stage-wise Riccati
recursion it replaces: same Newton step (primal, costates) up to rounding.  The algebra the device code follows (riccati_blk_mfma.hpp).

    python tools/probes/block2_newton_model.py
"""
import numpy as np

NX, NU, NZ = 8, 2, 10


def random_problem(rng, N):
    A = rng.normal(size=(N, NX, NX)) * 0.2 + np.eye(NX) * 0.7
    B = rng.normal(size=(N, NX, NU))
    rb = rng.normal(size=(N, NX)) * 0.1
    H = np.zeros((N + 1, NZ, NZ)); gt = rng.normal(size=(N + 1, NZ))
    for k in range(N + 1):
        Mx = rng.normal(size=(NZ, NZ)); H[k] = Mx @ Mx.T / NZ + np.diag(np.exp(rng.uniform(-2, 8, NZ)))      # barrier weights up to e^8
    H[N, 8:, :] = 0; H[N, :, 8:] = 0; gt[N, 8:] = 0
    return A, B, rb, H, gt


def stagewise(A, B, rb, H, gt):
    """Reference: Riccati recursion stage by stage (oracle/ihm2_oracle_qp.c: riccati_backward / riccati_forward)."""
    N = A.shape[0]
    P = np.zeros((N + 1, NX, NX)); p = np.zeros((N + 1, NX)); K = np.zeros((N, NU, NX)); kff = np.zeros((N, NU))
    P[N] = H[N, :8, :8]; p[N] = gt[N, :8]
    for k in range(N - 1, -1, -1):
        AB = np.hstack([A[k], B[k]])
        G = H[k] + AB.T @ P[k + 1] @ AB
        g = gt[k] + AB.T @ (P[k + 1] @ rb[k] + p[k + 1])
        Gi = np.linalg.inv(G[8:, 8:])
        K[k] = Gi @ G[8:, :8]; kff[k] = Gi @ g[8:]
        P[k] = G[:8, :8] - G[:8, 8:] @ K[k]; P[k] = 0.5 * (P[k] + P[k].T)
        p[k] = g[:8] - K[k].T @ g[8:]
    dz = np.zeros((N + 1, NZ)); dpi = np.zeros((N + 1, NX))
    dx = np.zeros(NX)
    for k in range(N):
        du = -K[k] @ dx - kff[k]
        dz[k, :8] = dx; dz[k, 8:] = du
        dpi[k] = P[k] @ dx + p[k]
        dx = A[k] @ dx + B[k] @ du + rb[k]
    dz[N, :8] = dx; dpi[N] = P[N] @ dx + p[N]
    return dz, dpi


def block2(A, B, rb, H, gt):
    """Stages paired (a, b) = (2m, 2m + 1); x_b is eliminated: block variables (x_a, u_a, u_b), block dynamics x_{a+2} = At x_a + Bt (u_a, u_b) + rt."""
    N = A.shape[0]; assert N % 2 == 0
    Nb = N // 2
    P = np.zeros((Nb + 1, NX, NX)); p = np.zeros((Nb + 1, NX)); K = np.zeros((Nb, 4, NX)); kff = np.zeros((Nb, 4))
    Mt = np.zeros((Nb, NX, NX)); ct = np.zeros((Nb, NX))
    P[Nb] = H[N, :8, :8]; p[Nb] = gt[N, :8]
    for m in range(Nb - 1, -1, -1):
        a, b = 2 * m, 2 * m + 1
        Y = np.hstack([A[a], B[a]])                                   # x_b = Y z_a + rb_a
        At = A[b] @ A[a]; Bt = np.hstack([A[b] @ B[a], B[b]]); rt = A[b] @ rb[a] + rb[b]
        Hb = H[b]
        Hblk = np.zeros((12, 12)); gblk = np.zeros(12)
        Hblk[:10, :10] = H[a] + Y.T @ Hb[:8, :8] @ Y
        Hblk[:10, 10:] = Y.T @ Hb[:8, 8:]; Hblk[10:, :10] = Hblk[:10, 10:].T
        Hblk[10:, 10:] = Hb[8:, 8:]
        gblk[:10] = gt[a] + Y.T @ (gt[b, :8] + Hb[:8, :8] @ rb[a])
        gblk[10:] = gt[b, 8:] + Hb[8:, :8] @ rb[a]
        Psi = np.hstack([At, Bt])                                     # 8 x 12
        G = Hblk + Psi.T @ P[m + 1] @ Psi
        g = gblk + Psi.T @ (P[m + 1] @ rt + p[m + 1])
        Gi = np.linalg.inv(G[8:, 8:])
        K[m] = Gi @ G[8:, :8]; kff[m] = Gi @ g[8:]
        P[m] = G[:8, :8] - G[:8, 8:] @ K[m]; P[m] = 0.5 * (P[m] + P[m].T)
        p[m] = g[:8] - K[m].T @ g[8:]
        Mt[m] = At - Bt @ K[m]; ct[m] = rt - Bt @ kff[m]
    dz = np.zeros((N + 1, NZ)); dpi = np.zeros((N + 1, NX))
    dx = np.zeros(NX)
    for m in range(Nb):                                               # the sequential part: Nb stages
        dz[2 * m, :8] = dx
        dx = Mt[m] @ dx + ct[m]
    dz[N, :8] = dx
    for m in range(Nb):                                               # everything below is parallel over the blocks
        a, b = 2 * m, 2 * m + 1
        dU = -K[m] @ dz[a, :8] - kff[m]
        dz[a, 8:] = dU[:2]; dz[b, 8:] = dU[2:]
        dz[b, :8] = A[a] @ dz[a, :8] + B[a] @ dU[:2] + rb[a]
        dpi[a] = P[m] @ dz[a, :8] + p[m]
    dpi[N] = P[Nb] @ dz[N, :8] + p[Nb]
    for m in range(Nb):
        b = 2 * m + 1
        dpi[b] = H[b, :8] @ dz[b] + gt[b, :8] + A[b].T @ dpi[b + 1]   # adjoint step: stationarity of the Newton system at the interior stage
    return dz, dpi


if __name__ == "__main__":
    rng = np.random.default_rng(3)
    worst = 0.0
    for trial in range(20):
        prob = random_problem(rng, 40)
        z1, p1 = stagewise(*prob); z2, p2 = block2(*prob)
        ez = np.max(np.abs(z1 - z2)) / np.max(np.abs(z1)); ep = np.max(np.abs(p1 - p2)) / np.max(np.abs(p1))
        worst = max(worst, ez, ep)
    print("block-2 Newton step against the stage-wise recursion: worst relative deviation over 20 random problems %.2e" % worst)
    assert worst < 1e-9
