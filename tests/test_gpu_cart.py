"""GPU parity, Cartesian side of the ROS stack: plants kin6 / dyn6 / speed switch and the
Cartesian -> Frenet projection, through the C ABI, against the oracle; and the ROS loop (project -> NMPC -> Cartesian plant)."""
import numpy as np
import pytest
from conftest import make_ocp, sample_x0

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cart(track):
    from ihm2_amd.closed_loop_sim import frenet_to_cartesian
    from ihm2_amd.solver import BatchedOcpSolver
    from oracle import oracle as orc

    B = 100
    s = BatchedOcpSolver(make_ocp(), B, track.s_ref, track.kappa_ref)
    s.set_track_geometry(track.X_ref, track.Y_ref, track.phi_ref)
    xf = sample_x0(track, B, seed=21)
    xf[:, 4] = 0.05 * xf[:, 3] * np.sign(xf[:, 5])
    xc = frenet_to_cartesian(track, xf)
    return dict(s=s, orc=orc, B=B, xf=xf, xc=xc)


def test_cartesian_plants_match_oracle(cart):
    s, orc, B, xc = cart["s"], cart["orc"], cart["B"], cart["xc"].copy()
    rng = np.random.default_rng(0)
    u = np.stack([rng.uniform(-400, 400, B), rng.uniform(-0.4, 0.4, B)], 1)
    xc[:10, 3] = np.linspace(0.02, 2.9, 10); xc[:10, 4] = 0.0; xc[:5, 6] = -50.0; u[:5, 0] = -300.0      # slow cars, some braking
    for model in (orc.MODEL_KIN6, orc.MODEL_DYN6, -3):
        got = s.sim_step_cart(xc, u, model=model, M_sim=10, dt_sim=0.01, n_steps=1)
        want = orc.sim_step_cart(xc, u, model, 10, dt=0.01)
        assert np.max(np.abs(got - want) / (1.0 + np.abs(want))) < 1e-11          # tolerance 1e-11 relative
    # five plant steps under a constant input = five single steps
    got5 = s.sim_step_cart(xc, u, model=-3, M_sim=10, dt_sim=0.01, n_steps=5)
    want5 = xc
    for _ in range(5):
        want5 = orc.sim_step_cart(want5, u, -3, 10, dt=0.01)
    assert np.max(np.abs(got5 - want5) / (1.0 + np.abs(want5))) < 1e-10
    assert np.any(got5[:10, 3] == 0.0)          # a braking slow car was stopped, not reversed


def test_projection_matches_oracle_and_inverts_the_embedding(cart, track):
    s, orc, B, xf, xc = cart["s"], cart["orc"], cart["B"], cart["xf"], cart["xc"]
    rng = np.random.default_rng(1)
    sg = xf[:, 0] + rng.uniform(-1.0, 1.0, B)
    got, sg_next = s.project(xc, sg)
    want, sg_want = orc.cart_to_frenet(track.s_ref, track.X_ref, track.Y_ref, track.phi_ref, xc, sg)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-11)
    np.testing.assert_allclose(sg_next, sg_want, rtol=0, atol=1e-11)
    # the Frenet state the Cartesian state was built from comes back (up to the polyline approximation of the spline)
    assert np.max(np.abs(got[:, 0] - xf[:, 0])) < 0.05 and np.max(np.abs(got[:, 1] - xf[:, 1])) < 0.02
    assert np.max(np.abs(got[:, 2] - xf[:, 2])) < 0.03
    np.testing.assert_array_equal(got[:, 3:], xc[:, 3:])


def test_ros_loop_with_cartesian_plant(track):
    """project -> NMPC (fkin6) -> Cartesian kin6/dyn6 plant at 100 Hz, from rest: the cars drive down the track."""
    from ihm2_amd.closed_loop_sim import CartesianSimulator, SimModelVariant, frenet_to_cartesian, run_closed_loop_cartesian
    from ihm2_amd.controller import IHM2Controller

    B = 32
    ctrl = IHM2Controller(track.s_ref, track.kappa_ref, batch_size=B, terminal_bounds="stage")
    sim = CartesianSimulator(ctrl, track, SimModelVariant.CART_KIN6_DYN6)
    xf0 = np.zeros((B, 8)); xf0[:, 0] = -6.0; xf0[1:, 1] = np.linspace(-0.3, 0.3, B - 1)
    res = run_closed_loop_cartesian(ctrl, sim, frenet_to_cartesian(track, xf0), xf0[:, 0].copy(), 80, lap_length=track.lap_length)
    assert res.alive.sum() >= 0.9 * B
    assert np.median(res.progress) > 15.0 and res.x[-1, :, 3].max() > 8.0
    assert np.abs(res.x[:, :, 1]).max() < 2.0 + 1e-6
    ctrl.solver.free()
