import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table
from oracle import oracle as orc
names = ("fsds_competition_1", "fsds_competition_2", "fsds_competition_3", "fsds_default", "acceleration", "skidpad", "short_skidpad")
plans = [track_table(n) for n in names]
N = 40; B = 12 * len(plans)
for model in ("fdyn6u",):
    ocp = make_ocp(model=model)
    s_ref = np.stack([p.s_ref for p in plans]); k_ref = np.stack([p.kappa_ref for p in plans])
    tid = (np.arange(B) % len(plans)).astype(np.int32)
    s = BatchedOcpSolver(ocp, B, s_ref, k_ref, track_id=tid)
    P = orc.OracleProblem(ocp.flatten().as_dict(s_ref, k_ref))
    x0 = np.zeros((B, 8))
    for t, p in enumerate(plans):
        sel = tid == t
        x0[sel] = sample_x0(p, int(sel.sum()), seed=300 + t)
    x0[:, 3] = np.clip(x0[:, 3], 4.0, 12.0)
    s.set_x0(x0); s.init_guess()
    x, u = s.get_x(), s.get_u()
    s.linearize()
    A, Bm, b = s.get_linearization()
    Ao, Bo, bo = P.linearize(x, u, track_id=tid)
    d = np.abs(b - bo).max(axis=(1, 2))
    print(model, "max |b - bo| per instance:", np.round(d, 12))
    i = int(np.argmax(d)); k = int(np.argmax(np.abs(b[i] - bo[i]).max(axis=1)))
    print("worst instance", i, "track", names[tid[i]], "interval", k, "x", x[i, k], "u", u[i, k], "b gpu", b[i, k], "b orc", bo[i, k])
