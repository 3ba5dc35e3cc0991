"""Diagnostic: the fdyn6 RTI step of tests/test_gpu_parity.py against the oracle, with a library variant chosen by IHM2_LIB."""
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

N = 40
track = track_table("fsds_competition_1")
for model in sys.argv[1:] or ["fdyn6", "fdyn6u"]:
    B = 40
    ocp = make_ocp(model=model)
    s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, B, seed=77); x0[:, 3] = np.linspace(4.0, 14.0, B)
    s.set_x0(x0); s.init_guess()
    x, u = s.get_x(), s.get_u()
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(N)[None] / N
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
    s.set_yref(yref); s.set_yref_e(yref_e); s.set_multipliers(None, None)
    st = s.solve(); it = s.get_qp_iter()
    out = P.rti_step(x, u, x0, yref, yref_e)
    print(model, "lib", _lib.LIB_PATH)
    print(" status gpu ", st)
    print(" status orc ", out["status"])
    print(" iter   gpu ", it)
    print(" iter   orc ", out.get("qp_iter"))
    print(" equal share", np.mean(st == out["status"]))
