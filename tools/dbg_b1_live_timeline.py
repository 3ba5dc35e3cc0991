"""Diagnostic: ONE car with the live solver options (SQP x 2, MERIT_BACKTRACKING, IRK) stepping in a loop -- for a kernel-trace timeline:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_b1l -- python3 tools/dbg_b1_live_timeline.py
    python3 tools/dbg_b1_live_timeline.py --analyse gpurun_out/prof_b1l
"""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

if "--analyse" in sys.argv:
    d = sys.argv[sys.argv.index("--analyse") + 1]
    f = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0])
                   for r in csv.DictReader(open(f))), key=lambda t: t[0])
    rows = rows[len(rows) // 2:]
    calls, cur = [], []
    for r in rows:
        if r[2].startswith("k_wrap_lap") and cur:
            calls.append(cur); cur = []
        cur.append(r)
    calls = calls[5:-1]
    c = calls[len(calls) // 2]
    t0 = c[0][0]
    for s, e, n in c: print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n}")
    span = [(c[-1][1] - c[0][0]) / 1e3 for c in calls]; busy = [sum(e - s for s, e, _ in c) / 1e3 for c in calls]
    print(f"first start to last end {np.mean(span):.1f} us, kernels busy {np.mean(busy):.1f} us, calls {len(calls)}")
    sys.exit(0)

from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table
track = track_table("fsds_competition_1")
s = BatchedOcpSolver(make_ocp(nlp_solver_type="SQP", nlp_solver_max_iter=2, globalization="MERIT_BACKTRACKING", integrator_type="IRK", sim_method_num_steps=1), 1, track.s_ref, track.kappa_ref)
s.set_x0(sample_x0(track, 1, seed=3)); s.init_guess(); s.set_lap_wrap(True)
for _ in range(300):
    s.step(40.0, model=0, M_sim=25); s.get_u0()
