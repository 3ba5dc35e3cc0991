"""Diagnostic: closed-loop step latency of small batches through ihm2mpc_compute_control / ihm2mpc_step with the four-wave QP kernel
(k_qp_block) on and off (IHM2MPC_BLOCK_QP=0): p50 / p99 per call for B = 1, 8, 64, 256."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table

track = track_table("fsds_competition_1")
# "live": the solver options the reference runs (python/main.py:227-238: SQP x 2, MERIT_BACKTRACKING, IRK) instead of SQP_RTI + RK4 x 25
LIVE = dict(nlp_solver_type="SQP", nlp_solver_max_iter=2, globalization="MERIT_BACKTRACKING", integrator_type="IRK", sim_method_num_steps=1)
opts = LIVE if "live" in sys.argv[1:] else {}
for B in (1, 8, 64, 256):
    s = BatchedOcpSolver(make_ocp(**opts), B, track.s_ref, track.kappa_ref)
    x0 = sample_x0(track, B, seed=3)
    s.set_x0(x0); s.init_guess(); s.set_lap_wrap(True)
    for _ in range(20):
        s.step(40.0, model=0, M_sim=25); s.get_u0()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); s.step(40.0, model=0, M_sim=25); u0 = s.get_u0(); ts.append(time.perf_counter() - t0)
    tm = s.get_timings()
    print(f"B {B:4d}: step p50 {np.percentile(ts, 50) * 1e3:.3f} ms p99 {np.percentile(ts, 99) * 1e3:.3f} ms  last qp_ms {tm['qp_ms']:.3f} linearize_ms {tm['linearize_ms']:.3f} "
          f"iterations {s.get_qp_iter().mean():.2f} ok {np.mean(np.isin(s.get_status(), (0, 2))):.3f}", flush=True)
    s.free()
