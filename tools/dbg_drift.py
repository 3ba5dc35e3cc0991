"""Diagnostic: per-step GPU-vs-oracle deviation of one long closed loop (the drift test's workload)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table
from oracle import oracle as orc
track = track_table("fsds_competition_1")
N = 40
B, M_sim, L = 48, 25, track.lap_length
ocp = make_ocp()
s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
x0 = sample_x0(track, B, seed=77)
s.set_x0(x0); s.init_guess(); s.set_lap_wrap(True)
x, u = s.get_x(), s.get_u()
s.prepare_step(40.0); st = s.solve()
yref, yref_e = orc.prepare_step(N, x0, 40.0, x, u)
out = P.rti_step(x, u, x0, yref, yref_e)
pi, lam = out["pi"], out["lam"]
xc = x0.copy()
rel = lambda a, b: np.max(np.abs(a - b) / (1 + np.abs(b)), axis=-1)
for chunk in range(6):
    h = s.run_steps(40.0, 25, model=0, M_sim=M_sim, u0_hist=True, x0_hist=True, status_hist=True, qp_iter_hist=True)
    for i in range(25):
        w = xc[:, 0] >= L
        xc[w, 0] -= L; x[w, :, 0] -= L
        xc = P.sim_step(xc, u[:, 0].copy(), 0, M_sim)
        yref, yref_e = orc.prepare_step(N, xc, 40.0, x, u)
        out = P.rti_step(x, u, xc, yref, yref_e, pi=pi, lam=lam)
        pi, lam = out["pi"], out["lam"]
        du = rel(h["u0"][i], u[:, 0]); dx = rel(h["x0"][i], xc)
        b = int(np.argmax(du))
        flips = np.flatnonzero(h["qp_iter"][i] != out["qp_iter"])
        print(f"step {chunk*25+i:3d} max du {du.max():.2e} (inst {b}, it gpu {h['qp_iter'][i][b]} orc {out['qp_iter'][b]}, st {h['status'][i][b]}/{out['status'][b]}, wrapped {bool(w[b])}) max dx {dx.max():.2e} flips {flips.tolist()}")
