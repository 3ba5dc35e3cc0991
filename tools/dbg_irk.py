import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table
from oracle import oracle as orc
track = track_table("fsds_competition_1")
B = 37
ocp = make_ocp(M=1, model="fkin6", integrator_type="IRK")
s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
x0 = sample_x0(track, B, seed=61); x0[:, 3] = np.linspace(4.0, 14.0, B)
s.set_x0(x0); s.init_guess()
x, u = s.get_x(), s.get_u()
s.linearize()
A, Bm, b = s.get_linearization()
Ao, Bo, bo = P.linearize(x, u)
bad = np.argwhere(np.isnan(Ao).any(axis=(2, 3)))
print("oracle NaN at", bad[:10], "gpu NaN", np.isnan(A).sum())
np.save(os.path.join(ROOT, "gpurun_out", "irk_x.npy"), x); np.save(os.path.join(ROOT, "gpurun_out", "irk_u.npy"), u)
for i, k in bad[:3]:
    print(i, k, repr(x[i, k]), repr(u[i, k]))
    print(orc.rk4_sens(0, x[i, k], u[i, k], track.s_ref, track.kappa_ref, 0.05, 1, integrator=1)[1])
ok = ~np.isnan(Ao).any(axis=(2, 3))
print("max dev where finite", np.max(np.abs(A[ok] - Ao[ok])), np.max(np.abs(b[ok] - bo[ok])))
import time
for name, kw in (("RK4x25", dict(M=25)), ("IRK", dict(M=1, integrator_type="IRK"))):
    s2 = BatchedOcpSolver(make_ocp(**kw), 1024, track.s_ref, track.kappa_ref)
    s2.set_x0(sample_x0(track, 1024)); s2.init_guess(); s2.linearize(); s2.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): s2.linearize()
    s2.synchronize(); print(name, (time.perf_counter() - t0) / 20 * 1e3, "ms")
