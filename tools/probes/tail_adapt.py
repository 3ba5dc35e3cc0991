"""Diagnostic: per-instance IPM iteration totals over a 20-step launch with the start parameters of an instance chosen from its previous solve's
iteration count (IHM2MPC_ADAPT=thr,mu0,tau0 in the experimental build)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ihm2_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "probes", "libihm2mpc_adapt.so")
from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table
track = track_table("fsds_competition_1")
B = 1024
x0 = sample_x0(track, B)
for spec in sys.argv[1:]:
    os.environ["IHM2MPC_ADAPT"] = spec
    s = BatchedOcpSolver(make_ocp(), B, track.s_ref, track.kappa_ref)
    s.set_x0(x0); s.init_guess(); s.set_lap_wrap(True)
    s.run_steps(40.0, 5, model=0, M_sim=25)
    h = s.run_steps(40.0, 20, model=0, M_sim=25, qp_iter_hist=True, status_hist=True)
    tot = h["qp_iter"].sum(axis=0)
    h2 = s.run_steps(40.0, 200, model=0, M_sim=25, qp_iter_hist=True, status_hist=True)
    print("adapt %-14s: 20 steps mean %.1f p99 %.0f max %d ok %.4f | next 200 steps: mean per solve %.3f ok %.4f" % (spec, tot.mean(), np.percentile(tot, 99), tot.max(), (h["status"] == 0).mean(),
          h2["qp_iter"].mean(), (h2["status"] == 0).mean()), flush=True)
    s.free()
