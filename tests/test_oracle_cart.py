"""Oracle, Cartesian side of the ROS stack: the kin6 / dyn6 plants against the NumPy mirror of
python/models.py:168-229, 310-452 (dyn6: the explicit form must solve the reference's IMPLICIT residual), the plant step
semantics of sim_node.cpp:197-257, and Track::project + the Frenet states of mpc_control_node.cpp:142-157 against an
independent NumPy restatement and against closed-form geometry."""
import numpy as np

from oracle import models_np as mnp
from oracle import oracle as orc

Z2 = np.zeros(2)


def _rand_cart(rng):
    x = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(-3, 3), rng.uniform(0.5, 20), rng.uniform(-1.5, 1.5),
                  rng.uniform(-1.5, 1.5), rng.uniform(-300, 300), rng.uniform(-0.4, 0.4)])
    u = np.array([rng.uniform(-400, 400), rng.uniform(-0.45, 0.45)])
    return x, u


def test_kin6_matches_numpy_mirror():
    rng = np.random.default_rng(0)
    for _ in range(200):
        x, u = _rand_cart(rng)
        f = orc.f(orc.MODEL_KIN6, x, u, Z2, Z2)
        np.testing.assert_allclose(f, mnp.kin6(x, u), rtol=1e-13, atol=1e-12)
    # the force model is fkin6's; only the kinematics and the yaw equation differ (python/models.py:226 vs :304)
    x, u = _rand_cart(rng)
    s_ref = np.array([-1e3, 1e3]); k0 = np.zeros(2)
    ff = orc.f(orc.MODEL_FKIN6, np.r_[0.0, 0.0, 0.0, x[3:]], u, s_ref, k0)
    fc = orc.f(orc.MODEL_KIN6, np.r_[0.0, 0.0, 0.0, x[3:]], u, Z2, Z2)
    np.testing.assert_allclose(fc[[3, 4, 6, 7]], ff[[3, 4, 6, 7]], rtol=1e-13, atol=1e-12)
    assert abs(fc[5] - fc[4] / mnp.l_R) < 1e-12


def test_dyn6_explicit_solves_reference_implicit_residual():
    rng = np.random.default_rng(1)
    scale = np.array([1, 1, 1, mnp.m * 10, mnp.m * 10, mnp.I_z * 10, 1e3, 1e2])
    for _ in range(200):
        x, u = _rand_cart(rng)
        xdot = orc.f(orc.MODEL_DYN6, x, u, Z2, Z2)
        assert np.max(np.abs(mnp.dyn6_residual(xdot, x, u)) / scale) < 1e-11


def test_dyn6_is_open_loop_stable_unlike_fdyn6_as_written():
    """The Cartesian dynamic model keeps every wheel on its own slip angle: no eigenvalue with a positive real part beyond
    rounding (X, Y, phi are integrators), where fdyn6 as written has +34 1/s (test_oracle_model.py)."""
    x = np.array([0.0, 0.0, 0.0, 10.0, 0.0, 0.0, 50.0, 0.0]); u = np.array([50.0, 0.0])
    _, J = orc.jac(orc.MODEL_DYN6, x, u, Z2, Z2)
    ev = np.linalg.eigvals(J[:, :8])
    assert ev.real.max() < 1e-6
    assert np.sum(ev.real < -5.0) >= 4          # lateral / yaw dynamics and the two actuators are well damped


def test_plant_step_switch_and_no_reversing():
    rng = np.random.default_rng(2)
    B = 12
    x = np.array([_rand_cart(rng)[0] for _ in range(B)]); u = np.array([_rand_cart(rng)[1] for _ in range(B)])
    x[:4, 3] = [0.5, 1.0, 2.0, 2.9]; x[:4, 4] = 0.0          # below v_dyn = 3: kinematic
    x[4:8, 3] = [3.1, 5.0, 8.0, 12.0]                         # above: dynamic
    xs = orc.sim_step_cart(x, u, -3, 10, dt=0.01)
    xk = orc.sim_step_cart(x, u, orc.MODEL_KIN6, 10, dt=0.01)
    xd = orc.sim_step_cart(x, u, orc.MODEL_DYN6, 10, dt=0.01)
    use_kin = np.hypot(x[:, 3], x[:, 4]) < 3.0
    ref = np.where(use_kin[:, None], xk, xd)
    stopped = (ref[:, 3] < 0.0) | ((ref[:, 6] <= 0.1) & (ref[:, 3] < 0.01))
    ref[stopped, 3:6] = 0.0
    np.testing.assert_array_equal(xs, ref)
    # braking at walking speed: the car stops instead of reversing
    x0 = np.array([[0, 0, 0, 0.05, 0, 0, -200.0, 0.0]]); u0 = np.array([[-400.0, 0.0]])
    x1 = orc.sim_step_cart(x0, u0, -3, 10, dt=0.01)
    assert np.all(x1[0, 3:6] == 0.0)
    # kin6 against scipy on one step (RK4 x 200: the torque actuator, t_T = 1 ms, needs sub-steps well below t_T for accuracy)
    from scipy.integrate import solve_ivp
    sol = solve_ivp(lambda t, y: mnp.kin6(y, u[5]), (0, 0.01), x[5], method="Radau", rtol=1e-12, atol=1e-14)
    xfine = orc.sim_step_cart(x[5:6], u[5:6], orc.MODEL_KIN6, 200, dt=0.01)[0]
    assert np.max(np.abs(sol.y[:, -1] - xfine) / (1 + np.abs(xfine))) < 1e-8


def _wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def _project_np(s_ref, X_ref, Y_ref, phi_ref, X, Y, s_guess, s_tol):
    """Independent NumPy restatement of tracks.cpp:183-288."""
    n = len(s_ref)
    lo = max(np.searchsorted(s_ref, max(s_guess - s_tol, s_ref[0]), side="right") - 1, 0)
    up = np.searchsorted(s_ref, min(s_guess + s_tol, s_ref[-1]), side="right") - 1
    lo = lo - 1 if lo > 0 else lo
    up = up + 1 if up < n - 1 else up
    P = np.stack([X_ref[lo:up + 1], Y_ref[lo:up + 1]], 1)
    car = np.array([X, Y])
    i = int(np.argmin(((P - car) ** 2).sum(1)))
    ip, inx = (i - 1) % len(P), (i + 1) % len(P)
    ang = lambda a, b, c: abs(_wrap(np.arctan2(c[1] - b[1], c[0] - b[0]) - np.arctan2(a[1] - b[1], a[0] - b[0])))
    if ang(P[i], car, P[ip]) > ang(P[i], car, P[inx]):
        a, b, sa, sb = P[ip], P[i], s_ref[lo + ip], s_ref[lo + i]
    else:
        a, b, sa, sb = P[i], P[inx], s_ref[lo + i], s_ref[lo + inx]
    lam = np.dot(car - a, b - a) / np.dot(b - a, b - a)
    s = sa + lam * (sb - sa)
    ind = min(lo + i, n - 2)
    phi = phi_ref[ind] + (phi_ref[ind + 1] - phi_ref[ind]) / (s_ref[ind + 1] - s_ref[ind]) * (s - s_ref[ind])
    return s, a[0] + lam * (b[0] - a[0]), a[1] + lam * (b[1] - a[1]), phi


def test_projection_matches_numpy_restatement(track):
    rng = np.random.default_rng(3)
    for _ in range(300):
        s0 = rng.uniform(-300, 600)
        Xc, Yc = np.interp(s0, track.s_ref, track.X_ref), np.interp(s0, track.s_ref, track.Y_ref)
        X, Y = Xc + rng.uniform(-1.5, 1.5), Yc + rng.uniform(-1.5, 1.5)
        sg = s0 + rng.uniform(-1.0, 1.0)
        got = orc.project(track.s_ref, track.X_ref, track.Y_ref, track.phi_ref, X, Y, sg, 2.0)
        want = _project_np(track.s_ref, track.X_ref, track.Y_ref, track.phi_ref, X, Y, sg, 2.0)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)


def test_frenet_states_on_a_straight_and_on_a_circle():
    # straight track along +x: s = X, n = Y, psi = phi
    s_ref = np.linspace(-50, 100, 301); X_ref = s_ref.copy(); Y_ref = np.zeros_like(s_ref); phi_ref = np.zeros_like(s_ref)
    xc = np.array([[12.3, 0.7, 0.2, 5.0, 0.1, 0.0, 10.0, 0.01], [40.0, -1.1, -0.3 + 2 * np.pi, 8.0, 0, 0, 0, 0]])
    xf, sg = orc.cart_to_frenet(s_ref, X_ref, Y_ref, phi_ref, xc, np.array([12.0, 41.0]))
    np.testing.assert_allclose(xf[:, 0], [12.3, 40.0], atol=1e-12)
    np.testing.assert_allclose(xf[:, 1], [0.7, -1.1], atol=1e-12)
    np.testing.assert_allclose(xf[:, 2], [0.2, -0.3], atol=1e-12)
    np.testing.assert_array_equal(xf[:, 3:], xc[:, 3:])
    np.testing.assert_allclose(sg, np.fmod(xf[:, 0] + 0.05 * xc[:, 3], 50.0), atol=1e-12)      # lap length = -s_ref[0]
    # circle of radius R, counter-clockwise: a point at radius R - n has offset +n (left of the direction of travel)
    R, nk = 30.0, 2001
    s_ref = np.linspace(0, 2 * np.pi * R, nk); th = s_ref / R
    X_ref, Y_ref, phi_ref = R * np.cos(th), R * np.sin(th), th + np.pi / 2
    for th0, n in ((0.7, 0.9), (2.0, -1.2), (4.0, 0.3)):
        X, Y = (R - n) * np.cos(th0), (R - n) * np.sin(th0)
        xf, _ = orc.cart_to_frenet(s_ref, X_ref, Y_ref, phi_ref, np.array([[X, Y, th0 + np.pi / 2 + 0.1, 5, 0, 0, 0, 0.0]]), np.array([R * th0 + 0.5]))
        assert abs(xf[0, 0] - R * th0) < 2e-3 and abs(xf[0, 1] - n) < 2e-3 and abs(xf[0, 2] - 0.1) < 2e-3      # chord vs arc
