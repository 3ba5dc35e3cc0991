"""The IRK integrators on the GPU (kernels_irk.hip) against the oracle: the reference's live option for the shooting intervals
(python/main.py:234-236: IRK, 4 stages, 1 step -- Gauss-Legendre) and for its plants (python/main.py:395-400: Radau IIA x 100)."""
import time

import numpy as np
import pytest
from conftest import make_ocp, sample_x0

pytestmark = pytest.mark.gpu

N = 40


def _rel(a, b, floor=1.0):
    return float(np.max(np.abs(a - b) / (floor + np.abs(b))))


@pytest.mark.parametrize("model", ["fkin6", "fdyn6u", "fdyn6"])
@pytest.mark.parametrize("colloc", ["GAUSS_LEGENDRE", "GAUSS_RADAU_IIA"])
def test_irk_linearisation_matches_oracle(track, model, colloc):
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import oracle as orc

    B = 37       # B * N is not a multiple of 16: the last wavefront has idle quads
    ocp = make_ocp(M=1, model=model, integrator_type="IRK", collocation_type=colloc)
    s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, B, seed=61)
    x0[:, 3] = np.linspace(4.0, 14.0, B)
    s.set_x0(x0); s.init_guess()
    x, u = s.get_x(), s.get_u()
    s.linearize()
    A, Bm, b = s.get_linearization()
    Ao, Bo, bo = P.linearize(x, u)
    colscale = np.maximum(np.abs(Ao).max(axis=2, keepdims=True), 1e-30)
    assert np.max(np.abs(A - Ao) / colscale) < 1e-9            # tolerance 1e-9 relative to the column scale
    colscale = np.maximum(np.abs(Bo).max(axis=2, keepdims=True), 1e-30)
    assert np.max(np.abs(Bm - Bo) / colscale) < 1e-9
    assert np.max(np.abs(b - bo)) < 1e-10
    assert np.all(A[:, :, 3:, :3] == 0) and np.all(A[:, :, 6:, :6] == 0)        # structural zeros are exact


def test_irk_rti_steps_match_oracle(track):
    """Two RTI iterations with the live integrator option (IRK, Gauss-Legendre, 4 stages, 1 step per interval)."""
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import oracle as orc

    B = 80
    ocp = make_ocp(M=1, integrator_type="IRK")
    s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, B, seed=62)
    s.set_x0(x0); s.init_guess()
    x, u = s.get_x(), s.get_u()
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(N)[None] / N
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
    s.set_yref(yref); s.set_yref_e(yref_e); s.set_multipliers(None, None)
    pi = lam = None
    for it in range(2):
        st = s.solve()
        out = P.rti_step(x, u, x0, yref, yref_e, pi=pi, lam=lam)
        pi, lam = out["pi"], out["lam"]
        np.testing.assert_array_equal(st, out["status"])
        ok = st == 0
        assert ok.sum() >= 0.9 * B
        np.testing.assert_array_equal(s.get_qp_iter()[ok], out["qp_iter"][ok])
        assert _rel(s.get_x()[ok], x[ok]) < 1e-7 and _rel(s.get_u()[ok], u[ok]) < 1e-7      # tolerance 1e-7 relative (north star: 1e-5)


@pytest.mark.parametrize("plant", [0, 2, -2])
def test_irk_plant_step_matches_oracle(track, plant):
    """The reference's plant integrator: IRK Radau IIA, 4 stages, 100 steps over dt (python/main.py:395-400)."""
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import oracle as orc

    B = 50
    ocp = make_ocp(sim_integrator_type="IRK", sim_collocation_type="GAUSS_RADAU_IIA")
    s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, B, seed=63)
    x0[:, 3] = np.linspace(2.5, 14.0, B)          # spans both sides of the kinematic / dynamic switch
    u = np.stack([np.linspace(-200, 300, B), np.linspace(-0.3, 0.3, B)], 1)
    xn = s.sim_step(x0, u, model=plant, M_sim=100)
    if plant >= 0:
        xo = P.sim_step(x0, u, plant, 100, integrator=orc.INTEG_IRK_RADAU4)
    else:      # python/main.py:482-489: kinematic below the side-slip threshold, dynamic above
        from ihm2_amd.constants import l_R

        beta = np.arctan(0.5 * np.tan(x0[:, 7]))
        kin = (x0[:, 3] ** 2 + x0[:, 4] ** 2) * np.sin(beta) / l_R <= 3.0
        assert kin.any() and (~kin).any()
        xo = np.where(kin[:, None], P.sim_step(x0, u, 0, 100, integrator=orc.INTEG_IRK_RADAU4), P.sim_step(x0, u, 2, 100, integrator=orc.INTEG_IRK_RADAU4))
    assert _rel(xn, xo) < 1e-10


def test_irk_needs_a_fraction_of_the_rk4_work(track):
    """Linearisation time at B = 1024: IRK (3 Newton iterations x 4 stages + 4 = 16 model evaluations per interval) beside RK4 x 25
    (100 evaluations); printed for DESIGN.md.  Asserted on the dynamic model, where the gap is 4x (1.5 against 6.1 ms at B = 8192): the
    kinematic model's 0.135 against 0.19 ms is too close for a timing assertion on a shared box."""
    from ihm2_amd.solver import BatchedOcpSolver

    B, t = 1024, {}
    for name, kw in (("RK4x25", dict(M=25)), ("IRK GL4x1", dict(M=1, integrator_type="IRK")),
                     ("fdyn6u RK4x25", dict(M=25, model="fdyn6u")), ("fdyn6u IRK GL4x1", dict(M=1, integrator_type="IRK", model="fdyn6u"))):
        s = BatchedOcpSolver(make_ocp(**kw), B, track.s_ref, track.kappa_ref)
        s.set_x0(sample_x0(track, B)); s.init_guess()
        s.linearize(); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            s.linearize()
        s.synchronize()
        t[name] = (time.perf_counter() - t0) / 20 * 1e3
        s.free()
    print("linearisation of 1024 x 40 intervals [ms]:", t)
    assert t["fdyn6u IRK GL4x1"] < 0.7 * t["fdyn6u RK4x25"]


def test_sim_solver_object_replays_the_reference_plant_calls(track):
    """``generate_sim_solver`` + ``AcadosSimSolver.set / solve / get / simulate`` as ``python/main.py:395-435,476-502`` use them:
    three plant objects (fkin6, fdyn6, fdyn6u) with the reference's options, ``p`` set afterwards, the model switch of the loop."""
    from ihm2_amd import ocp as O
    from ihm2_amd.constants import l_R
    from ihm2_amd.sim import AcadosSimOpts, generate_sim_solver
    from oracle import oracle as orc

    dt = 0.05
    sim_opts = AcadosSimOpts()
    sim_opts.T = dt; sim_opts.num_stages = 4; sim_opts.num_steps = 100
    sim_opts.integrator_type = "IRK"; sim_opts.collocation_type = "GAUSS_RADAU_IIA"
    model = O.get_acados_model_from_explicit_dynamics("ihm2_fkin6", O.fkin6_model, 8, 2, 3000)
    model_fdyn6 = O.get_acados_model_from_implicit_dynamics("fdyn6", O.fdyn6u_model, 8, 2, 3000)
    sim_solver = generate_sim_solver(model, sim_opts, "generated", generate=True, build=True)
    sim_solver_fdyn6 = generate_sim_solver(model_fdyn6, sim_opts, "generated", generate=True, build=True)
    p = np.append(track.s_ref, track.kappa_ref)
    sim_solver.set("p", p); sim_solver_fdyn6.set("p", p)
    P = orc.OracleProblem(make_ocp().flatten().as_dict(track.s_ref, track.kappa_ref))
    x = np.zeros(8); x[0] = -6.0                   # python/main.py:438-441
    used = set()
    for i in range(25):
        u = np.array([400.0 if i < 18 else 100.0, 0.25 * np.sin(0.4 * i)])
        beta = np.arctan(0.5 * np.tan(x[7]))
        kin = (x[3] ** 2 + x[4] ** 2) * np.sin(beta) / l_R <= 3.0      # python/main.py:482-489
        used.add(bool(kin))
        solver = sim_solver if kin else sim_solver_fdyn6
        xn = solver.simulate(x, u)
        assert xn.shape == (8,) and solver.get("CPUtime") > 0.0
        xo = P.sim_step(x[None], u[None], 0 if kin else 2, 100, integrator=orc.INTEG_IRK_RADAU4)[0]
        assert _rel(xn, xo) < 1e-10, i
        x = xn
    assert used == {True, False}                   # both plants were exercised
    # set / solve / get, batched
    B = 33
    sb = generate_sim_solver(model, sim_opts, "generated", batch_size=B)
    sb.set("p", p); sb.set("T", dt)
    xs = sample_x0(track, B, seed=9); us = np.stack([np.linspace(-100, 300, B), np.linspace(-0.2, 0.2, B)], 1)
    sb.set("x", xs); sb.set("u", us)
    assert sb.solve() == 0
    assert _rel(sb.get("x"), P.sim_step(xs, us, 0, 100, integrator=orc.INTEG_IRK_RADAU4)) < 1e-10
    for s_ in (sim_solver, sim_solver_fdyn6, sb):
        s_.free()
