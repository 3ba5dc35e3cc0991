#!/usr/bin/env python3
"""Time the linearisation kernel alone (B instances x N intervals, RK4 x M or IRK) and check it against the oracle on a sample.
usage: tools/bench_linearize.py [--model fdyn6u] [--batch 8192] [--lib path] [--integrator ERK|IRK]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="fdyn6u"); ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--lib", default=None); ap.add_argument("--integrator", default="ERK"); ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--check", type=int, default=64)
ap.add_argument("--cols", action="store_true", help="fkin6 / ERK: the column-parallel kernel (one sensitivity column per wavefront) at this batch size")
args = ap.parse_args()
if args.cols: os.environ["IHM2MPC_LINEARIZE_COLS"] = "1"
from ihm2_amd import _lib
if args.lib: _lib.LIB_PATH = os.path.abspath(args.lib)
from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table

track = track_table("fsds_competition_1")
kw = dict(integrator_type="IRK", sim_method_num_steps=1) if args.integrator == "IRK" else {}
ocp = make_ocp(model=args.model, M=1 if args.integrator == "IRK" else 25, **kw)
s = BatchedOcpSolver(ocp, args.batch, track.s_ref, track.kappa_ref)
x0 = sample_x0(track, args.batch, seed=5)
if args.model != "fkin6": x0[:, 3] = np.linspace(6.0, 14.0, args.batch)
s.set_x0(x0); s.init_guess()
s.linearize(); s.synchronize()
t0 = time.perf_counter()
for _ in range(args.reps): s.linearize()
s.synchronize()
ms = (time.perf_counter() - t0) / args.reps * 1e3
out = {"model": args.model, "batch": args.batch, "integrator": args.integrator, "column_parallel": bool(args.cols), "linearize_ms": ms}
if args.check:
    from oracle import oracle as orc
    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    n = args.check
    A, B, b = s.get_linearization()
    Ao, Bo, bo = P.linearize(s.get_x()[:n], s.get_u()[:n])
    rel = lambda a, c: float(np.max(np.abs(a - c)) / max(1.0, np.max(np.abs(c))))
    out["max_rel_dev_vs_oracle"] = max(rel(A[:n], Ao), rel(B[:n], Bo), rel(b[:n], bo))
print(out)
