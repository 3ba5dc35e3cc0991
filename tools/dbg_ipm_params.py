"""Sweep of the interior-point start parameters (qp_mu0, qp_tau0) on the bench workload: mean IPM iterations, statuses, step time."""
import sys, time, itertools
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table
track = track_table("fsds_competition_1")
B = 1024
x0 = sample_x0(track, B)
GRID = itertools.product((0.003, 0.01, 0.03, 0.1, 0.3, 1.0), (0.1, 0.3, 1.0))
if "--fine" in sys.argv:
    GRID = itertools.product((0.05, 0.1, 0.2), (0.5, 1.0, 2.0, 4.0))
for mu0, tau0 in GRID:
    ocp = make_ocp(qp_mu0=mu0, qp_tau0=tau0)
    s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
    s.set_x0(x0); s.init_guess(); s.set_lap_wrap(True)
    s.run_steps(40.0, 20, model=0, M_sim=25)
    s.synchronize(); t0 = time.perf_counter()
    h = s.run_steps(40.0, 200, model=0, M_sim=25, status_hist=True, qp_iter_hist=True)
    s.synchronize(); el = time.perf_counter() - t0
    it = h["qp_iter"]; tot20 = it[:20].sum(axis=0)
    w = it.reshape(10, 20, B).sum(axis=1)            # ten windows of 20 steps: the launch time of a 20-step run follows the window's slowest instance
    win = float(np.mean(w.max(axis=1)))
    print(f"mu0 {mu0:6.3f} tau0 {tau0:4.1f}: iterations mean {it.mean():5.2f} p99 {np.percentile(it, 99):4.0f} max {it.max():3d}  status0 {np.mean(h['status'] == 0):.4f}  "
          f"{B * 200 / el / 1e3:7.1f} k solves/s  20-step max/mean {tot20.max() / tot20.mean():.3f}  slowest instance per 20-step window (mean of 10) {win:.1f} iterations", flush=True)
    s.free()
