"""SURVEY.md section 8f N2 remainder on the GPU: the 15-state plant ``fdyn10`` (python/models.py:609-801) through
``ihm2mpc_sim_step_dyn10`` against the oracle, and the ``AcadosSimSolver``-shaped object on it (python/main.py:490-502)."""
import numpy as np
import pytest

from conftest import make_ocp
from test_oracle_dyn10 import _states

pytestmark = pytest.mark.gpu


def test_dyn10_plant_steps_match_oracle(track):
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import oracle as orc

    B = 192
    s = BatchedOcpSolver(make_ocp(), B, track.s_ref, track.kappa_ref)
    x, u = _states(track, B, seed=21)
    for M in (100, 25):
        xn = s.sim_step_dyn10(x, u, M_sim=M)
        xo = orc.sim_step_dyn10(x, u, track.s_ref, track.kappa_ref, M, dt=0.05)
        assert np.all(np.isfinite(xn))
        assert np.max(np.abs(xn - xo) / np.maximum(1.0, np.abs(xo))) < 1e-9           # tolerance 1e-9 relative
    with pytest.raises(Exception, match="unstable"):
        s.sim_step_dyn10(x, u, M_sim=5)
    # ten plant steps in a row stay together (the model is stiff in the wheel slips: errors are not amplified)
    xg, xc = x.copy(), x.copy()
    for _ in range(10):
        xg = s.sim_step_dyn10(xg, u, M_sim=100); xc = orc.sim_step_dyn10(xc, u, track.s_ref, track.kappa_ref, 100, dt=0.05)
    ok = np.all(np.isfinite(xc), axis=1)
    assert ok.mean() > 0.9 and np.max(np.abs(xg[ok] - xc[ok]) / np.maximum(1.0, np.abs(xc[ok]))) < 1e-7
    s.free()


def test_sim_solver_object_on_the_dyn10_plant(track):
    """python/main.py:395-428,490-502 with the DYN10 plant: generate_sim_solver(model, opts, ...), set("p"), simulate(x, u)."""
    from ihm2_amd import ocp as O
    from ihm2_amd.sim import AcadosSimOpts, generate_sim_solver
    from oracle import oracle as orc

    model = O.get_acados_model_from_implicit_dynamics("ihm2_fdyn10", O.fdyn10_model, 15, 5, 2 * track.s_ref.size)
    assert (model.nx, model.nu) == (15, 5)
    opts = AcadosSimOpts(T=0.05, num_stages=4, num_steps=100, integrator_type="ERK")
    sim = generate_sim_solver(model, opts, "/tmp/unused", generate=False, build=False)
    sim.set("p", np.append(track.s_ref, track.kappa_ref))
    x, u = _states(track, 1, seed=4)
    xn = sim.simulate(x[0], u[0])
    assert xn.shape == (15,)
    xo = orc.sim_step_dyn10(x, u, track.s_ref, track.kappa_ref, 100, dt=0.05)[0]
    assert np.max(np.abs(xn - xo) / np.maximum(1.0, np.abs(xo))) < 1e-9
    with pytest.raises(ValueError, match="plant"):
        O.get_acados_ocp(model, 10, 2.0, 31.0, 500.0, 0.5, 1e6, 1.0).flatten()      # fdyn10 has no OCP
    sim.free()


def _radau_solver(track, B):
    from ihm2_amd.solver import BatchedOcpSolver

    return BatchedOcpSolver(make_ocp(sim_integrator_type="IRK", sim_collocation_type="GAUSS_RADAU_IIA"), B, track.s_ref, track.kappa_ref)


def test_dyn10_radau_plant_from_rest_matches_oracle(track):
    """the reference's start x0 = (-6, 0, ..., 0) (python/main.py:438-441) under the reference's plant integrator
    (python/main.py:395-400: IRK, GAUSS_RADAU_IIA, 4 stages, 100 steps).  A batch of cars at rest with different torques and steering
    commands, plant step after plant step on the GPU and in the oracle: no NaN, states equal to 1e-9 relative (the oracle agrees with
    scipy's adaptive Radau to 1e-9 from rest, tests/test_oracle_dyn10.py)."""
    from oracle import oracle as orc

    B = 16
    s = _radau_solver(track, B)
    x = np.zeros((B, 15)); x[:, 0] = -6.0
    u = np.zeros((B, 5)); u[:, :4] = np.linspace(10.0, 125.0, B)[:, None]; u[:, 4] = np.linspace(-0.2, 0.2, B)
    xg, xo = x.copy(), x.copy()
    for k in range(12):
        xg = s.sim_step_dyn10(xg, u, M_sim=100)
        xo = orc.sim_step_dyn10_irk(xo, u, track.s_ref, track.kappa_ref)
        assert np.all(np.isfinite(xg)) and np.all(np.isfinite(xo))
        assert np.max(np.abs(xg - xo) / (1.0 + np.abs(xo))) < 1e-9, k              # tolerance 1e-9 relative
    assert xg[-1, 3] > 3.0          # 0.6 s at full torque: the last car is under way
    # 100 plant steps (5 s of driving from rest) without a NaN
    for k in range(88):
        xg = s.sim_step_dyn10(xg, u, M_sim=100)
    assert np.all(np.isfinite(xg))
    mid = B // 2                    # nearly straight (u_delta = 0.013), 71 N m per wheel: well under way after 5 s
    assert xg[mid, 3] > 5.0 and xg[mid, 0] > 5.0
    s.free()


def test_dyn10_radau_plant_on_moving_cars_matches_oracle_and_rk4(track):
    from oracle import oracle as orc

    B = 96
    s = _radau_solver(track, B)
    x, u = _states(track, B, seed=33)
    xg = s.sim_step_dyn10(x, u, M_sim=100)
    xo = orc.sim_step_dyn10_irk(x, u, track.s_ref, track.kappa_ref)
    assert np.max(np.abs(xg - xo) / (1.0 + np.abs(xo))) < 1e-9                 # tolerance 1e-9 relative
    xk = orc.sim_step_dyn10(x, u, track.s_ref, track.kappa_ref, 400, dt=0.05)
    assert np.max(np.abs(xg - xk) / (1.0 + np.abs(xk))) < 1e-8                 # the two integrators agree on a moving car
    s.free()


def test_sim_solver_object_replays_the_reference_dyn10_loop_from_rest(track):
    """python/main.py:395-441,490-502: AcadosSimOpts(T = dt, num_stages = 4, num_steps = 100, IRK, GAUSS_RADAU_IIA), generate_sim_solver on
    fdyn10, set("p"), x = zeros(15) with x[0] = -6, then simulate(x, [u_T / 4] * 4 + [u_delta]) step after step."""
    from ihm2_amd import ocp as O
    from ihm2_amd.sim import AcadosSimOpts, generate_sim_solver
    from oracle import oracle as orc

    model = O.get_acados_model_from_implicit_dynamics("ihm2_fdyn10", O.fdyn10_model, 15, 5, 2 * track.s_ref.size)
    opts = AcadosSimOpts(T=0.05, num_stages=4, num_steps=100, integrator_type="IRK", collocation_type="GAUSS_RADAU_IIA")
    sim = generate_sim_solver(model, opts, "/tmp/unused", generate=False, build=False)
    sim.set("p", np.append(track.s_ref, track.kappa_ref))
    x = np.zeros(15); x[0] = -6.0
    u_T, u_delta = 400.0, 0.02
    u = np.array([0.25 * u_T] * 4 + [u_delta])
    xo = x.copy()
    for i in range(10):
        x = sim.simulate(x, u)
        assert x.shape == (15,) and not np.any(np.isnan(x))            # python/main.py:503-504
        xo = orc.sim_step_dyn10_irk(xo, u, track.s_ref, track.kappa_ref)[0]
    assert np.max(np.abs(x - xo) / (1.0 + np.abs(xo))) < 1e-9
    assert x[3] > 1.5
    sim.free()


def test_dyn10_mil_loop_from_rest_under_the_stanley_controller(track):
    """python/main.py:438-517 with sim_model_variant = DYN10 as the reference runs it: x0 = (-6, 0, ..., 0), the Stanley controller on
    (n, psi, v_x) with a speed reference, a quarter of the torque on every wheel, Radau IIA x 100 plant steps -- six cars with different
    speed references for 80 control periods (4 s): nobody is lost, every car settles near its reference speed and stays on the track."""
    from ihm2_amd.closed_loop_sim import run_closed_loop_dyn10

    v_ref = np.array([3.0, 4.0, 5.0, 6.0, 8.0, 10.0])
    res = run_closed_loop_dyn10(track, n_steps=80, batch_size=6, v_x_ref=v_ref)
    assert res.x.shape == (81, 6, 15) and np.all(np.isfinite(res.x)) and res.alive.all()
    assert np.all(res.x[0, :, 0] == -6.0) and np.all(res.x[0, :, 3:] == 0.0)                 # from rest
    assert np.all(np.abs(res.x[-1, :, 3] - v_ref) < 0.7)                                     # the PI speed loop has settled (k_P = 90: 0.5 m/s droop against the drag)
    assert np.all(np.abs(res.x[-1, :, 1]) < 0.5) and np.all(res.x[-1, :, 0] > 0.0)           # on the track, past the start line
    assert np.all(np.abs(res.u[..., 0]) <= 500.0) and np.all(np.abs(res.u[..., 1]) <= 0.5)
    # wheels roll: omega R_w ~ v_x for every wheel once under way
    assert np.all(np.abs(res.x[-1, :, 6:10] * 0.20809 / res.x[-1, :, 3:4] - 1.0) < 0.1)
