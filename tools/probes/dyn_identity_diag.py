"""Where the persistent loop of the dynamic OCP models and launches per step part: plant state, linearisation records, iterate after ONE step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["IHM2MPC_BLOCK_QP"] = "0"
from conftest import make_ocp, sample_x0
from ihm2_amd.track import track_table
from ihm2_amd.solver import BatchedOcpSolver
track = track_table("fsds_competition_1")
B, M_sim = 70, 30
x0 = sample_x0(track, B, seed=91)
for model, plant in (("fdyn6u", 2), ("fdyn6u", 0)):
    out = []
    for persistent in (False, True):
        s = BatchedOcpSolver(make_ocp(model=model), B, track.s_ref, track.kappa_ref)
        s.set_lap_wrap(True); s.set_x0(x0); s.init_guess()
        s.step(40.0, model=plant, M_sim=M_sim)
        if persistent: s.run_steps(40.0, 1, model=plant, M_sim=M_sim)
        else: s.step(40.0, model=plant, M_sim=M_sim)
        A, Bm, b = s.get_linearization()
        out.append(dict(x0=s.get_x0(), A=A, B=Bm, b=b, x=s.get_x(), u=s.get_u(), it=s.get_qp_iter()))
        s.free()
    for k in out[0]:
        a, c = out[0][k], out[1][k]
        d = np.abs(a - c)
        print(model, plant, k, "equal" if np.array_equal(a, c) else f"DIFFER max abs {d.max():.3e} rel {np.max(d / np.maximum(1e-300, np.abs(c))):.3e} count {int((d > 0).sum())} of {d.size}", flush=True)
    a, c = out[0]["A"], out[1]["A"]
    pos = np.argwhere(a != c)
    if not len(pos): continue
    from collections import Counter
    print("A positions (row, col) -> count:", sorted(Counter((int(p[-2]), int(p[-1])) for p in pos).items()))
    i = tuple(pos[0]); print("sample", i, repr(a[i]), repr(c[i]))
    a, c = out[0]["B"], out[1]["B"]
    pos = np.argwhere(a != c)
    print("B positions (row, col) -> count:", sorted(Counter((int(p[-2]), int(p[-1])) for p in pos).items()))
    i = tuple(pos[0]); print("sample", i, repr(a[i]), repr(c[i]))
