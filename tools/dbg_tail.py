"""Diagnostic: IPM iterations per instance over a 20-step persistent launch (the launch ends with its slowest instance: DESIGN.md section 7 (f))."""
import sys, os, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table
track = track_table("fsds_competition_1")
B = 1024
ocp = make_ocp()
s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
x0 = sample_x0(track, B)
s.set_x0(x0); s.init_guess(); s.set_lap_wrap(True)
s.run_steps(40.0, 5, model=0, M_sim=25)
h = s.run_steps(40.0, 20, model=0, M_sim=25, qp_iter_hist=True, status_hist=True)
it = h["qp_iter"]  # (20, B)
tot = it.sum(axis=0)
print("per-instance total iterations over 20 steps: mean %.1f p50 %.0f p90 %.0f p99 %.0f max %d" % (tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max()))
print("per-step: mean %.2f max %d ; hist of per-solve iterations" % (it.mean(), it.max()), np.bincount(it.ravel()))
top = np.argsort(tot)[-8:]
print("slowest instances:", tot[top], "their per-step iterations:\n", it[:, top].T)
x = s.get_x0()[top]
print("their states (s,n,psi,vx,vy,r,T,delta):\n", np.round(x, 3))
# cost model: step time ~ lin 0.2 + 0.075*it
t = (0.2 * 20 + 0.075 * tot)
print("modelled ms: mean %.2f max %.2f ratio %.3f" % (t.mean(), t.max(), t.max() / t.mean()))
