import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def track():
    from ihm2_amd.track import track_table

    return track_table("fsds_competition_1")


def make_ocp(N=40, M=25, model="fkin6", n_max=2.0, **opts):
    from ihm2_amd import ocp as O

    fn = {"fkin6": O.fkin6_model, "fdyn6": O.fdyn6_model, "fdyn6u": O.fdyn6u_model}[model]
    mdl = O.get_acados_model_from_explicit_dynamics("ihm2_" + model, fn, 8, 2, 3000)
    ocp = O.get_acados_ocp(mdl, N, n_max, 31.0, 500.0, 0.5, 1e6, 1.0)
    ocp.cost.W, ocp.cost.W_e = O.default_weights()
    ocp.solver_options.tf = N * 0.05
    ocp.solver_options.sim_method_num_steps = M
    for k, v in opts.items():
        setattr(ocp.solver_options, k, v)
    return ocp


def sample_x0(track, B, seed=20240607):
    """Synthetic initial states of SURVEY.md section 8d."""
    from ihm2_amd.constants import l_R

    rng = np.random.default_rng(seed)
    s = rng.uniform(0, track.lap_length, B)
    n = rng.uniform(-0.5, 0.5, B)
    psi = rng.uniform(-0.1, 0.1, B)
    vx = rng.uniform(2, 15, B)
    kap = np.interp(s, track.s_ref, track.kappa_ref)
    T = rng.uniform(-100, 300, B)
    delta = np.arctan(2 * np.tan(np.arcsin(np.clip(kap * l_R, -0.9, 0.9))))
    return np.stack([s, n, psi, vx, 0 * s, vx * kap, T, delta], 1)


def random_state(rng):
    x = np.array([rng.uniform(0, 300), rng.uniform(-1, 1), rng.uniform(-0.3, 0.3), rng.uniform(0.5, 20),
                  rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-300, 300), rng.uniform(-0.4, 0.4)])
    u = np.array([rng.uniform(-400, 400), rng.uniform(-0.45, 0.45)])
    return x, u
