"""Diagnostic: per-instance IPM iteration totals over a 20-step launch for several start parameters (mu0, tau0): would a per-instance choice shorten the tail?"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table
track = track_table("fsds_competition_1")
B = 1024
x0 = sample_x0(track, B)
res = {}
for mu0 in (0.03, 0.1, 0.3):
    for tau0 in (0.3, 1.0, 3.0):
        s = BatchedOcpSolver(make_ocp(qp_mu0=mu0, qp_tau0=tau0), B, track.s_ref, track.kappa_ref)
        s.set_x0(x0); s.init_guess(); s.set_lap_wrap(True)
        s.run_steps(40.0, 5, model=0, M_sim=25)
        h = s.run_steps(40.0, 20, model=0, M_sim=25, qp_iter_hist=True, status_hist=True)
        tot = h["qp_iter"].sum(axis=0)
        res[(mu0, tau0)] = (tot, h["qp_iter"], (h["status"] == 0).mean())
        print("mu0 %.2f tau0 %.1f: mean %.1f p99 %.0f max %d ok %.4f" % (mu0, tau0, tot.mean(), np.percentile(tot, 99), tot.max(), res[(mu0, tau0)][2]), flush=True)
        s.free()
allt = np.stack([v[0] for v in res.values()])
best = allt.min(axis=0)
print("per-instance best of the nine: mean %.1f p99 %.0f max %d" % (best.mean(), np.percentile(best, 99), best.max()))
ref = res[(0.1, 1.0)][0]
slow = np.argsort(ref)[-20:]
print("slowest 20 under (0.1, 1.0):", ref[slow])
for k, v in res.items():
    print(k, v[0][slow])
# per-SOLVE correlation: iterations of a solve against the same instance's previous solve (is "slow" persistent?)
it = res[(0.1, 1.0)][1]
print("corr(it[t], it[t+1]) = %.3f" % np.corrcoef(it[:-1].ravel(), it[1:].ravel())[0, 1])
