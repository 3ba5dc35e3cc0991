"""BASELINE.json configs[0]: a single instance of the kinematic bicycle NMPC, N = 20 (python/generate_all.py:95,108: Nf = 20,
tf = 1.0), skidpad track, SQP-RTI -- the reference's own CPU-runnable case.  On the CPU the oracle drives the loop; on the GPU
the same loop runs through the AcadosOcpSolver-shaped view of a batch of one and must give the oracle's controls."""
import numpy as np
import pytest
from conftest import make_ocp

N0 = 20
S_TARGET = 6.0        # reference ramp: 6 m over the 1 s horizon (the skidpad circles have a 9 m radius; python/main.py uses 40 m over 2 s on the FSDS tracks)


def _problem():
    from ihm2_amd.track import track_table

    track = track_table("skidpad")
    ocp = make_ocp(N=N0)                 # tf = N * 0.05 = 1.0, M = 25
    x0 = np.array([[12.0, 0.05, 0.01, 4.0, 0.0, 0.0, 60.0, 0.0]])
    x0[0, 5] = x0[0, 3] * np.interp(x0[0, 0], track.s_ref, track.kappa_ref)
    return track, ocp, x0


def _oracle_loop(track, ocp, x0, steps):
    from oracle import oracle as orc

    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x = np.zeros((1, N0 + 1, 8)); u = np.zeros((1, N0, 2))
    x[:] = x0[:, None]; x[0, :, 0] = x0[0, 0] + x0[0, 3] * 0.05 * np.arange(N0 + 1)      # constant-speed guess along the centre line
    pi = lam = None
    xc = x0.copy()
    us, xs, sts = [], [xc.copy()], []
    for _ in range(steps):
        yref, yref_e = orc.prepare_step(N0, xc, S_TARGET, x, u)
        out = P.rti_step(x, u, xc, yref, yref_e, pi=pi, lam=lam)
        pi, lam = out["pi"], out["lam"]
        sts.append(int(out["status"][0])); us.append(u[0, 0].copy())
        xc = P.sim_step(xc, u[:, 0].copy(), 0, 25)
        xs.append(xc.copy())
    return np.array(us), np.array(xs)[:, 0], np.array(sts)


def test_config0_oracle_loop_drives_the_skidpad():
    track, ocp, x0 = _problem()
    us, xs, sts = _oracle_loop(track, ocp, x0, 60)
    assert np.all(sts == 0)
    assert xs[-1, 0] > xs[0, 0] + 5.0                      # 3 s of driving: the car moves along the centre line
    # plumbing, not tuning: with the reference's weights (q_delta = 100 against q_n = 1, python/main.py:193-210) and a 1 s
    # horizon the controller steers reluctantly into the 9 m circle; the hard bound |n| <= n_max = 2 is what must hold
    assert np.max(np.abs(xs[:, 1])) <= 2.0 + 1e-6
    assert np.all(np.abs(us[:, 0]) <= 500.0 + 1e-9) and np.all(np.abs(us[:, 1]) <= 0.5 + 1e-9)
    assert np.all(np.isfinite(xs))


@pytest.mark.gpu
def test_config0_gpu_single_instance_matches_the_oracle_loop():
    from ihm2_amd.solver import BatchedOcpSolver

    track, ocp, x0 = _problem()
    steps = 10
    us_o, xs_o, sts_o = _oracle_loop(track, ocp, x0, steps)
    batch = BatchedOcpSolver(ocp, 1, track.s_ref, track.kappa_ref)
    solver = batch[0]                                       # the AcadosOcpSolver-shaped view (python/main.py:297-334 call sequence)
    xg = np.zeros((N0 + 1, 8)); xg[:] = x0[0]; xg[:, 0] = x0[0, 0] + x0[0, 3] * 0.05 * np.arange(N0 + 1)
    for j in range(N0 + 1):
        solver.set(j, "x", xg[j])
    for j in range(N0):
        solver.set(j, "u", np.zeros(2))
    xc = x0[0].copy()
    x_pred = xg.copy(); u_pred = np.zeros((N0, 2))
    for i in range(steps):
        solver.set(0, "lbx", xc); solver.set(0, "ubx", xc)
        for j in range(N0):                                # reference ramp, python/main.py:303-314
            yr = np.zeros(12); yr[0] = xc[0] + S_TARGET * j / N0
            solver.set(j, "yref", yr)
        yr = np.zeros(8); yr[0] = xc[0] + S_TARGET
        solver.set(N0, "yref", yr)
        for j in range(N0 - 1):                            # shift, python/main.py:317-322
            solver.set(j, "x", x_pred[j + 1]); solver.set(j, "u", u_pred[j + 1])
        solver.set(N0 - 1, "x", x_pred[N0]); solver.set(N0 - 1, "u", np.zeros(2)); solver.set(N0, "x", x_pred[N0])
        status = solver.solve()
        assert status == sts_o[i] == 0
        x_pred = np.array([solver.get(j, "x") for j in range(N0 + 1)]); u_pred = np.array([solver.get(j, "u") for j in range(N0)])
        assert np.max(np.abs(u_pred[0] - us_o[i]) / (1.0 + np.abs(us_o[i]))) < 1e-6      # tolerance 1e-6 relative (north star 1e-5)
        xc = batch.sim_step(xc[None], u_pred[:1], model=0, M_sim=25)[0]
        assert np.max(np.abs(xc - xs_o[i + 1]) / (1.0 + np.abs(xs_o[i + 1]))) < 1e-6
    batch.free()
