"""Self-certifying checks of the oracle's Riccati interior-point QP solver: convex-QP KKT conditions
(necessary and sufficient since H_k > 0), agreement with scipy on small cases, failure reporting."""
import numpy as np
import pytest
from conftest import make_ocp, sample_x0
from scipy.optimize import minimize

from oracle import oracle as orc

NX, NU, NZ, NC = 8, 2, 10, 14


def random_qp(rng, N, with_ineq=True):
    H = np.zeros((N + 1, NZ, NZ)); g = rng.normal(size=(N + 1, NZ))
    for k in range(N + 1):
        Mx = rng.normal(size=(NZ, NZ))
        H[k] = Mx @ Mx.T / NZ + 0.5 * np.eye(NZ)
    H[N, 8:, :] = 0; H[N, :, 8:] = 0; H[N, 8, 8] = H[N, 9, 9] = 1.0; g[N, 8:] = 0
    A = rng.normal(size=(N, NX, NX)) * 0.2 + np.eye(NX) * 0.6
    Bm = rng.normal(size=(N, NX, NU))
    b = rng.normal(size=(N, NX)) * 0.1
    dx0 = rng.normal(size=NX) * 0.3
    R = np.zeros((N + 1, NC, NZ)); dl = np.full((N + 1, NC), -np.inf); du = np.full((N + 1, NC), np.inf)
    if with_ineq:
        # bounds are drawn around a simulated trajectory so that the QP is feasible by construction
        zf = np.zeros((N + 1, NZ)); zf[0, :NX] = dx0
        for k in range(N):
            zf[k, NX:] = rng.uniform(-0.3, 0.3, NU)
            zf[k + 1, :NX] = A[k] @ zf[k, :NX] + Bm[k] @ zf[k, NX:] + b[k]
        for k in range(N + 1):
            for i in range(NX):
                R[k, i, i] = 1.0
            if k >= 1:
                for i in rng.choice(NX, 3, replace=False):
                    dl[k, i], du[k, i] = zf[k, i] - rng.uniform(0.05, 1.0), zf[k, i] + rng.uniform(0.05, 1.0)
            if k < N:
                R[k, 8, 8] = R[k, 9, 9] = 1.0
                dl[k, 8:10], du[k, 8:10] = -rng.uniform(0.35, 1.0, 2), rng.uniform(0.35, 1.0, 2)
                R[k, 10] = rng.normal(size=NZ) * (rng.random(NZ) < 0.4)
                du[k, 10] = R[k, 10] @ zf[k] + rng.uniform(0.1, 2.0)          # one-sided general row
    return dict(H=H, g=g, A=A, Bm=Bm, b=b, dx0=dx0, R=R, dl=dl, du=du)


def kkt_report(qp, sol):
    """inf-norms of the four KKT residual groups of the QP at the returned point."""
    N = qp["A"].shape[0]
    z, pi, lam = sol["dz"], sol["pi"], sol["lam"]
    lam_l, lam_u = lam[:, :NC], lam[:, NC:]
    stat = 0.0; eq = np.max(np.abs(z[0, :NX] - qp["dx0"])); ineq = 0.0; comp = 0.0
    for k in range(N + 1):
        r = qp["H"][k] @ z[k] + qp["g"][k] - qp["R"][k].T @ (lam_l[k] - lam_u[k])
        if k < N:
            AB = np.hstack([qp["A"][k], qp["Bm"][k]])
            r += AB.T @ pi[k + 1]
            eq = max(eq, np.max(np.abs(AB @ z[k] + qp["b"][k] - z[k + 1, :NX])))
        r[:NX] -= pi[k]
        lo, hi = (NX if k == 0 else 0), (NZ if k < N else NX)
        stat = max(stat, np.max(np.abs(r[lo:hi])))
        Rz = qp["R"][k] @ z[k]
        for c in range(NC):
            if np.isfinite(qp["dl"][k, c]):
                ineq = max(ineq, qp["dl"][k, c] - Rz[c]); comp = max(comp, abs(lam_l[k, c] * (Rz[c] - qp["dl"][k, c])))
            if np.isfinite(qp["du"][k, c]):
                ineq = max(ineq, Rz[c] - qp["du"][k, c]); comp = max(comp, abs(lam_u[k, c] * (qp["du"][k, c] - Rz[c])))
    return stat, eq, ineq, comp, float(lam.min())


def test_unconstrained_qp_is_solved_in_one_newton_step():
    qp = random_qp(np.random.default_rng(0), 10, with_ineq=False)
    sol = orc.qp_solve(**qp, tol=1e-10)
    assert sol["status"] == 0 and sol["iters"] == 1
    stat, eq, ineq, comp, _ = kkt_report(qp, sol)
    assert stat < 1e-10 and eq < 1e-10


@pytest.mark.parametrize("seed", range(6))
def test_random_qp_satisfies_kkt(seed):
    qp = random_qp(np.random.default_rng(100 + seed), 12)
    sol = orc.qp_solve(**qp, tol=1e-7, iter_max=60)
    assert sol["status"] == 0
    stat, eq, ineq, comp, lam_min = kkt_report(qp, sol)
    sg = max(1.0, np.abs(qp["g"]).max())
    assert stat < 1.01e-7 * sg and eq < 1.01e-7 and ineq < 1.01e-7 and comp < 1.01e-7 * sg and lam_min >= 0.0


def test_small_qp_matches_scipy():
    N = 3
    qp = random_qp(np.random.default_rng(7), N)
    sol = orc.qp_solve(**qp, tol=1e-8, iter_max=60)
    assert sol["status"] == 0
    # eliminate states: decision = (u_0..u_{N-1}); objective and constraints via forward simulation
    def rollout(uv):
        u = uv.reshape(N, NU); z = np.zeros((N + 1, NZ)); z[0, :NX] = qp["dx0"]
        for k in range(N):
            z[k, NX:] = u[k]
            z[k + 1, :NX] = qp["A"][k] @ z[k, :NX] + qp["Bm"][k] @ u[k] + qp["b"][k]
        return z
    def obj(uv):
        z = rollout(uv)
        return sum(0.5 * z[k] @ qp["H"][k] @ z[k] + qp["g"][k] @ z[k] for k in range(N + 1))
    cons = []
    for k in range(N + 1):
        for c in range(NC):
            if np.isfinite(qp["dl"][k, c]):
                cons.append({"type": "ineq", "fun": lambda uv, k=k, c=c: qp["R"][k, c] @ rollout(uv)[k] - qp["dl"][k, c]})
            if np.isfinite(qp["du"][k, c]):
                cons.append({"type": "ineq", "fun": lambda uv, k=k, c=c: qp["du"][k, c] - qp["R"][k, c] @ rollout(uv)[k]})
    ref = minimize(obj, np.zeros(N * NU), constraints=cons, method="SLSQP", options={"ftol": 1e-14, "maxiter": 500})
    assert ref.success
    u_ipm = sol["dz"][:N, NX:].ravel()
    assert obj(u_ipm) <= ref.fun + 1e-8 * (1 + abs(ref.fun))
    np.testing.assert_allclose(u_ipm, ref.x, atol=2e-5)


def test_nmpc_qp_kkt_and_iteration_count(track):
    """The QP the RTI step assembles on the bicycle problem: solved to tolerance in a bounded number
    of interior-point iterations."""
    ocp = make_ocp()
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, 8)
    N = 40
    iters = []
    for b in range(8):
        x = np.zeros((N + 1, 8)); u = np.zeros((N, 2)); x[0] = x0[b]
        for k in range(N):
            u[k] = [x0[b, 6], x0[b, 7]]
            x[k + 1] = orc.rk4(0, x[k], u[k], track.s_ref, track.kappa_ref, 0.05, 25)
        x[:, 1] = np.clip(x[:, 1], -1.5, 1.5)      # keep the linearisation point inside the track
        yref = np.zeros((N, 12)); yref[:, 0] = x0[b, 0] + 40.0 * np.arange(N) / N
        yref_e = np.zeros(8); yref_e[0] = x0[b, 0] + 40.0
        qp = P.build_qp(x, u, x0[b], yref, yref_e)
        sol = orc.qp_solve(**qp, tol=1e-6, iter_max=40)
        assert sol["status"] == 0
        stat, eq, ineq, comp, lam_min = kkt_report(qp, sol)
        sg = max(1.0, np.abs(qp["g"]).max()); sb = max(1.0, np.abs(qp["b"]).max(), np.abs(qp["dx0"]).max())
        assert stat <= 1e-6 * sg * 1.01 and eq <= 1e-6 * sb * 1.01 and ineq <= 1e-6 * sb * 1.01 and comp <= 1.01e-6 * sg
        assert lam_min >= 0
        iters.append(sol["iters"])
    assert max(iters) <= 25


def test_infeasible_qp_is_reported_not_nan():
    qp = random_qp(np.random.default_rng(3), 6)
    qp["dl"][3, 8], qp["du"][3, 8] = 0.5, 0.6
    qp["dl"][3, 9], qp["du"][3, 9] = -0.1, 0.1
    qp["R"][3, 10] = 0; qp["R"][3, 10, 8] = 1.0; qp["R"][3, 10, 9] = 1.0
    qp["dl"][3, 10], qp["du"][3, 10] = -5.0, -4.0          # u0 + u1 <= -4 contradicts the boxes
    sol = orc.qp_solve(**qp, tol=1e-6, iter_max=40)
    assert sol["status"] in (2, 4)
    assert np.all(np.isfinite(sol["dz"]))


def test_soft_constraints_match_explicit_slack_formulation():
    """A QP whose hard version is infeasible: softened sides (L1 + L2 slack penalty, as
    old/generate_acaods_interface.py:380-395) against SLSQP with explicit slack variables."""
    N = 3
    rng = np.random.default_rng(21)
    qp = random_qp(rng, N)
    # make the state box of stage 2 impossible to meet, then soften exactly those sides
    k, c = 2, 1
    qp["R"][k, c] = 0; qp["R"][k, c, c] = 1.0
    zf = np.zeros((N + 1, NZ)); zf[0, :NX] = qp["dx0"]
    for kk in range(N):
        zf[kk + 1, :NX] = qp["A"][kk] @ zf[kk, :NX] + qp["b"][kk]
    qp["dl"][k, c], qp["du"][k, c] = zf[k, c] + 30.0, zf[k, c] + 31.0      # unreachable with |u| <= 1
    for kk in range(N):
        qp["dl"][kk, 10], qp["du"][kk, 10] = -np.inf, np.inf                   # keep the test about the softened box
    hard = orc.qp_solve(**qp, tol=1e-8, iter_max=60)
    assert hard["status"] in (2, 4)
    soft_z = np.zeros((N + 1, 2 * NC)); soft_Z = np.full((N + 1, 2 * NC), -1.0)
    for side in (c, NC + c):
        soft_z[k, side], soft_Z[k, side] = 100.0, 100.0
    sol = orc.qp_solve(**qp, tol=1e-8, iter_max=80, soft_z=soft_z, soft_Z=soft_Z)
    assert sol["status"] == 0
    assert sol["sl"][k, c] > 1.0 and sol["sl"][k, NC + c] < 1e-6           # only the lower side is violated
    assert np.all(sol["sl"] >= 0)

    def rollout(uv):
        u = uv.reshape(N, NU); z = np.zeros((N + 1, NZ)); z[0, :NX] = qp["dx0"]
        for kk in range(N):
            z[kk, NX:] = u[kk]
            z[kk + 1, :NX] = qp["A"][kk] @ z[kk, :NX] + qp["Bm"][kk] @ u[kk] + qp["b"][kk]
        return z

    def obj(v):
        z = rollout(v[:-2]); sl_, su_ = v[-2], v[-1]
        return (sum(0.5 * z[kk] @ qp["H"][kk] @ z[kk] + qp["g"][kk] @ z[kk] for kk in range(N + 1))
                + 100.0 * (sl_ + su_) + 50.0 * (sl_ ** 2 + su_ ** 2))

    cons = [{"type": "ineq", "fun": lambda v: v[-2]}, {"type": "ineq", "fun": lambda v: v[-1]}]
    for kk in range(N + 1):
        for cc in range(NC):
            if np.isfinite(qp["dl"][kk, cc]):
                extra = (lambda v: v[-2]) if (kk, cc) == (k, c) else (lambda v: 0.0)
                cons.append({"type": "ineq", "fun": lambda v, kk=kk, cc=cc, extra=extra: qp["R"][kk, cc] @ rollout(v[:-2])[kk] + extra(v) - qp["dl"][kk, cc]})
            if np.isfinite(qp["du"][kk, cc]):
                extra = (lambda v: v[-1]) if (kk, cc) == (k, c) else (lambda v: 0.0)
                cons.append({"type": "ineq", "fun": lambda v, kk=kk, cc=cc, extra=extra: qp["du"][kk, cc] - qp["R"][kk, cc] @ rollout(v[:-2])[kk] + extra(v)})
    # KKT of the slack block: Z s + z - lam - lam_s = 0, lam_s >= 0, lam_s s = 0
    lam_soft = sol["lam"][k, c]
    assert sol["sl"][k, c] > 0 and lam_soft == pytest.approx(100.0 * sol["sl"][k, c] + 100.0, rel=1e-6)
    assert sol["lam"][k, NC + c] <= 100.0 + 1e-6                              # inactive side: lam <= z
    stat, eq, ineq, comp, lam_min = kkt_report(dict(qp, dl=np.where(soft_Z[:, :NC] >= 0, -np.inf, qp["dl"]),
                                                    du=np.where(soft_Z[:, NC:] >= 0, np.inf, qp["du"])), sol)
    sg = max(1.0, np.abs(qp["g"]).max())
    assert stat < 1e-7 * sg and eq < 1e-7 and ineq < 1e-7 and lam_min >= 0
    # the violated constraint holds with its slack: R z + s >= dl
    assert qp["R"][k, c] @ sol["dz"][k] + sol["sl"][k, c] >= qp["dl"][k, c] - 1e-7
    # SLSQP started at the IPM point cannot improve the explicit-slack objective
    v_ipm = np.concatenate([sol["dz"][:N, NX:].ravel(), [sol["sl"][k, c], sol["sl"][k, NC + c]]])
    ref = minimize(obj, v_ipm, constraints=cons, method="SLSQP", options={"ftol": 1e-14, "maxiter": 200})
    assert ref.fun >= obj(v_ipm) - 1e-6 * (1 + abs(obj(v_ipm)))
    np.testing.assert_allclose(ref.x, v_ipm, rtol=1e-3, atol=1e-3)


def test_residuals_reported_at_the_iteration_limit_are_formed_from_the_data(track):
    """A QP stopped by its iteration limit reports -- and is accepted or rejected on -- residuals formed from A, B, R at the
    returned point, not the values the iteration carried along between two evaluations.  Late limits are the telling ones:
    with large barrier weights the carried stationarity residual keeps shrinking (2e-14 after 14 iterations) while the one formed from
    the data stalls at the rounding error of the Riccati solve (6e-9)."""
    ocp = make_ocp()
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, 4)
    N = 40
    for b in range(4):
        x = np.zeros((N + 1, 8)); u = np.zeros((N, 2)); x[0] = x0[b]
        for k in range(N):
            u[k] = [x0[b, 6], x0[b, 7]]
            x[k + 1] = orc.rk4(0, x[k], u[k], track.s_ref, track.kappa_ref, 0.05, 25)
        x[:, 1] = np.clip(x[:, 1], -1.5, 1.5)
        yref = np.zeros((N, 12)); yref[:, 0] = x0[b, 0] + 40.0 * np.arange(N) / N
        yref_e = np.zeros(8); yref_e[0] = x0[b, 0] + 40.0
        qp = P.build_qp(x, u, x0[b], yref, yref_e)
        sg = max(1.0, np.abs(qp["g"]).max())
        for iter_max in (3, 6, 12, 14):
            sol = orc.qp_solve(**qp, tol=1e-14, iter_max=iter_max, mu0=0.1, tau0=1.0)
            assert sol["status"] in (1, 4) and sol["iters"] == iter_max
            stat, eq, _, _, _ = kkt_report(qp, sol)
            # (two summation orders: equal to rounding, which is all that is left of the residual at the late limits)
            assert sol["stats"][0] == pytest.approx(stat, rel=0.02, abs=1e-14 * sg)
            assert sol["stats"][1] == pytest.approx(eq, rel=0.02, abs=2e-13)
            Rz = np.einsum("kcj,kj->kc", qp["R"], sol["dz"])
            t, lam = sol["t"], sol["lam"]
            fin = np.concatenate([np.isfinite(qp["dl"]), np.isfinite(qp["du"])], 1)
            assert sol["stats"][3] == pytest.approx(np.abs(lam * t)[fin].max(), rel=1e-9)
