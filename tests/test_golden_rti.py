"""Regression against tests/golden/rti_vectors.npz (vectors of this build's own algorithm, see make_rti_vectors.py):
the CPU oracle must reproduce them to rounding, the HIP path to the parity tolerance."""
import os

import numpy as np
import pytest
from conftest import make_ocp

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rti_vectors.npz"))
N = 40


def test_oracle_reproduces_the_golden_rti_step(track):
    from oracle import oracle as orc

    P = orc.OracleProblem(make_ocp().flatten().as_dict(track.s_ref, track.kappa_ref))
    x, u = G["x_in"].copy(), G["u_in"].copy()
    A, Bm, b = P.linearize(x, u)
    np.testing.assert_allclose(A, G["A"], rtol=0, atol=1e-12); np.testing.assert_allclose(b, G["b"], rtol=0, atol=1e-12)
    out = P.rti_step(x, u, G["x0"], G["yref"], G["yref_e"])
    np.testing.assert_array_equal(out["status"], G["status"]); np.testing.assert_array_equal(out["qp_iter"], G["qp_iter"])
    # same code, same compiler flags: agreement to rounding (1e-10 leaves room for another libm / FMA contraction)
    assert np.max(np.abs(x - G["x_out"]) / (1 + np.abs(G["x_out"]))) < 1e-10
    assert np.max(np.abs(u - G["u_out"]) / (1 + np.abs(G["u_out"]))) < 1e-10
    assert np.max(np.abs(out["res"] - G["res"]) / (1 + np.abs(G["res"]))) < 1e-10


@pytest.mark.gpu
def test_hip_path_reproduces_the_golden_rti_step(track):
    from ihm2_amd.solver import BatchedOcpSolver

    B = G["x0"].shape[0]
    s = BatchedOcpSolver(make_ocp(), B, track.s_ref, track.kappa_ref)
    s.set_x0(G["x0"]); s.set_x(G["x_in"]); s.set_u(G["u_in"]); s.set_yref(G["yref"]); s.set_yref_e(G["yref_e"]); s.set_multipliers(None, None)
    s.linearize()
    A, Bm, b = s.get_linearization()
    assert np.max(np.abs(A - G["A"]) / np.maximum(np.abs(G["A"]).max(axis=2, keepdims=True), 1e-30)) < 1e-10
    assert np.max(np.abs(b - G["b"])) < 1e-11
    status = s.solve()
    np.testing.assert_array_equal(status, G["status"]); np.testing.assert_array_equal(s.get_qp_iter(), G["qp_iter"])
    assert np.max(np.abs(s.get_x() - G["x_out"]) / (1 + np.abs(G["x_out"]))) < 1e-7       # tolerance 1e-7 relative (north star: 1e-5)
    assert np.max(np.abs(s.get_u() - G["u_out"]) / (1 + np.abs(G["u_out"]))) < 1e-7
    assert np.max(np.abs(s.get_residuals() - G["res"]) / (1 + np.abs(G["res"]))) < 1e-9
    pi, lam = s.get_multipliers()
    assert np.max(np.abs(lam - G["lam"])) / (1 + np.abs(G["lam"]).max()) < 1e-6
    s.free()
