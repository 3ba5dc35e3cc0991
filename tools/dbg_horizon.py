"""Diagnostic: tests/test_gpu_parity.py::test_other_horizons_match_oracle for one (N, B) with a library variant (IHM2_LIB)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ihm2_amd import _lib
if os.environ.get("IHM2_LIB"): _lib.LIB_PATH = os.environ["IHM2_LIB"]
from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table
from oracle import oracle as orc

Nh, B = int(sys.argv[1]), int(sys.argv[2])
track = track_table("fsds_competition_1")
ocp = make_ocp(N=Nh); ocp.solver_options.tf = Nh * 0.05
s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
x0 = sample_x0(track, B, seed=100 + Nh)
s.set_x0(x0); s.init_guess()
x, u = s.get_x(), s.get_u()
yref = np.zeros((B, Nh, 12)); yref[:, :, 0] = x0[:, 0:1] + Nh * np.arange(Nh)[None] / Nh
yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + Nh
s.set_yref(yref); s.set_yref_e(yref_e); s.set_multipliers(None, None)
pi = lam = None
for it in range(2):
    st = s.solve()
    out = P.rti_step(x, u, x0, yref, yref_e, pi=pi, lam=lam)
    pi, lam = out["pi"], out["lam"]
    bad = np.nonzero(st != out["status"])[0]
    print("it", it, "lib", _lib.LIB_PATH, "mismatch", bad, "gpu", st[bad], "orc", out["status"][bad], "iters gpu", s.get_qp_iter()[bad], "orc", out["qp_iter"][bad])
    fail = np.nonzero(out["status"] != 0)[0]
    print("   failing (oracle):", fail, "gpu iters", s.get_qp_iter()[fail], "orc iters", out["qp_iter"][fail], "gpu st", st[fail])
