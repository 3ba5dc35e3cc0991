"""SURVEY.md section 8a, A15 on the GPU: ``ihm2_amd/mpc.py::get_ipopt_solver`` (python/mpc.py:116-216) -- CasADi's nlpsol calling
convention over the SQP mode of libihm2mpc.so run to convergence -- against the oracle's converged SQP and against the KKT conditions of
the programme itself."""
import numpy as np
import pytest

from conftest import sample_x0

pytestmark = pytest.mark.gpu
NF, DT = 12, 0.05


def test_nlp_solver_returns_a_kkt_point_of_the_programme_and_matches_the_oracle(track):
    from ihm2_amd import mpc, ocp as O
    from oracle import oracle as orc

    Q = np.diag([0.0, 4.0, 2.0, 0.3, 0.1, 0.1, 1e-5, 1.0]); Q[1, 2] = Q[2, 1] = 0.5
    R = np.diag([1e-5, 2.0]); Qf = 3.0 * Q
    mb = mpc.ModelBounds(v_x_min=1.0, delta_max=0.3, delta_dot_max=2.0)
    solver, lbx, ubx, lbg, ubg = mpc.get_ipopt_solver(O.fkin6_model, NF, mb, DT, track.s_ref, track.kappa_ref, Q, R, Qf, max_iter=60, tol=1e-7)
    assert lbx.shape == ubx.shape == (8 * (NF + 1) + 2 * NF,) and lbg.shape == ubg.shape == (10 * NF,)
    assert np.all(lbg[:8 * NF] == 0) and np.all(ubg[8 * NF:] == np.tile([1e-3 * mb.T_dot_max, 0.02 * mb.delta_dot_max], NF))
    x0 = sample_x0(track, 3, seed=11)[1]
    x0[1], x0[2], x0[3] = 0.8, 0.12, 9.0
    lbx, ubx = lbx.copy(), ubx.copy()
    lbx[:8] = ubx[:8] = x0                                   # python/main.py:377-388: the measured state fixes the first column
    w0 = np.concatenate((np.tile(x0, NF + 1), np.tile(x0[6:8], NF)))
    with pytest.raises(ValueError):
        solver(x0=w0, lbx=lbx[:-1], ubx=ubx, lbg=lbg, ubg=ubg)
    sol = solver(x0=w0, lbx=lbx, ubx=ubx, lbg=lbg, ubg=ubg)
    st = solver.stats()
    assert st["success"] and st["return_status"] == "Solve_Succeeded" and 1 <= st["iter_count"] <= 60, st
    w, g = sol["x"], sol["g"]
    X, U = w[:8 * (NF + 1)].reshape(NF + 1, 8), w[8 * (NF + 1):].reshape(NF, 2)
    # primal feasibility
    assert np.array_equal(X[0], x0)
    assert np.all(w >= lbx - 1e-8) and np.all(w <= ubx + 1e-8) and np.all(g >= lbg - 1e-7) and np.all(g <= ubg + 1e-7)
    # the oracle's converged SQP on the same programme
    P = orc.OracleProblem(solver._batch.ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    xo = np.tile(x0, (1, NF + 1, 1)).copy(); uo = np.tile(x0[6:8], (1, NF, 1)).copy()
    out = P.sqp_solve(xo, uo, x0[None], np.zeros((1, NF, 12)), np.zeros((1, 8)), max_iter=60, tol=1e-7)
    assert out["status"][0] == 0
    assert np.max(np.abs(X - xo[0]) / np.maximum(1.0, np.abs(xo[0]))) < 1e-6 and np.max(np.abs(U - uo[0]) / np.maximum(1.0, np.abs(uo[0]))) < 1e-6   # tolerance 1e-6 relative
    f = float(np.einsum("ki,ij,kj->", X[:-1], Q, X[:-1]) + np.einsum("ki,ij,kj->", U, R, U) + X[-1] @ Qf @ X[-1])
    assert abs(sol["f"] - f) <= 1e-12 * max(1.0, abs(f))
    # stationarity of L = f + lam_g'g + lam_x'w in CasADi's convention, Jacobians of the shooting rows from the linearisation
    A, B, _ = solver._batch.get_linearization()
    A, B = A[0], B[0]
    lam_s, lam_l = sol["lam_g"][:8 * NF].reshape(NF, 8), sol["lam_g"][8 * NF:].reshape(NF, 2)
    gX = np.zeros((NF + 1, 8)); gU = np.zeros((NF, 2))
    for k in range(NF):
        gX[k] += 2.0 * Q @ X[k] + A[k].T @ lam_s[k]; gX[k + 1] -= lam_s[k]
        gX[k, 6:8] += lam_l[k]
        gU[k] += 2.0 * R @ U[k] + B[k].T @ lam_s[k] - lam_l[k]
    gX[NF] += 2.0 * Qf @ X[NF]
    grad = np.concatenate((gX.reshape(-1), gU.reshape(-1))) + sol["lam_x"]
    scale = max(1.0, np.max(np.abs(sol["lam_g"])), np.max(np.abs(sol["lam_x"])))
    assert np.max(np.abs(grad)) <= 1e-5 * scale, np.max(np.abs(grad))
    # complementarity / signs: a bound multiplier is positive only at an active upper bound, negative only at an active lower one
    lx = sol["lam_x"][8:]; wl, wu, ww = lbx[8:], ubx[8:], w[8:]
    tol = 1e-5 * scale
    assert np.all((lx <= tol) | (np.abs(ww - wu) <= 1e-5 * np.maximum(1.0, np.abs(wu))))
    assert np.all((lx >= -tol) | (np.abs(ww - wl) <= 1e-5 * np.maximum(1.0, np.abs(wl))))
    # a second call with other bounds reuses the handle (the steering limit tightened: the solution must respect it)
    ubx2, lbx2 = ubx.copy(), lbx.copy()
    ubx2[8 * (NF + 1) + 1::2] = 0.1; lbx2[8 * (NF + 1) + 1::2] = -0.1
    sol2 = solver(x0=sol["x"], lbx=lbx2, ubx=ubx2, lbg=lbg, ubg=ubg)
    assert solver.stats()["success"] and np.all(np.abs(sol2["x"][8 * (NF + 1) + 1::2]) <= 0.1 + 1e-8) and sol2["f"] >= sol["f"] - 1e-9
    solver.free()
