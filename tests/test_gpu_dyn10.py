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
    with pytest.raises(ValueError, match="RK4"):
        generate_sim_solver(model, AcadosSimOpts(T=0.05, num_steps=100, integrator_type="IRK", collocation_type="GAUSS_RADAU_IIA"), "/tmp/unused")
    with pytest.raises(ValueError, match="plant"):
        O.get_acados_ocp(model, 10, 2.0, 31.0, 500.0, 0.5, 1e6, 1.0).flatten()      # fdyn10 has no OCP
    sim.free()
