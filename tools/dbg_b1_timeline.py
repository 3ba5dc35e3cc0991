"""Diagnostic: one real-time controller (B = 1) calling ihm2mpc_compute_control in a loop -- for a rocprofv3 --kernel-trace timeline:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_b1 -- python3 tools/dbg_b1_timeline.py
    python3 tools/dbg_b1_timeline.py --analyse gpurun_out/prof_b1      # kernel durations and the gaps between them per call
"""
import os, sys, time, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

if "--analyse" in sys.argv:
    d = sys.argv[sys.argv.index("--analyse") + 1]
    f = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0])
                   for r in csv.DictReader(open(f))), key=lambda t: t[0])
    rows = rows[len(rows) // 2:]                       # steady state
    # a call = prepare, linearize_cols, qp
    calls, cur = [], []
    for r in rows:
        if r[2].startswith("k_prepare") and cur:
            calls.append(cur); cur = []
        cur.append(r)
    calls = [c for c in calls if len(c) >= 3][5:-1]
    dur = {}; gaps = []; span = []
    for c in calls:
        for s, e, n in c: dur.setdefault(n, []).append((e - s) / 1e3)
        gaps.append(sum(c[i + 1][0] - c[i][1] for i in range(len(c) - 1)) / 1e3); span.append((c[-1][1] - c[0][0]) / 1e3)
    for n, v in dur.items(): print(f"{n:32s} {np.mean(v):8.1f} us")
    print(f"gaps between the kernels of a call {np.mean(gaps):8.1f} us ; first start to last end {np.mean(span):8.1f} us ; calls {len(calls)}")
    period = np.diff([c[0][0] for c in calls]) / 1e3
    print(f"call period (host loop) {np.median(period):8.1f} us")
    sys.exit(0)

from conftest import make_ocp, sample_x0
from ihm2_amd.solver import BatchedOcpSolver
from ihm2_amd.track import track_table
track = track_table("fsds_competition_1")
s = BatchedOcpSolver(make_ocp(), 1, track.s_ref, track.kappa_ref)
x0 = sample_x0(track, 1, seed=3)
s.set_x0(x0); s.init_guess(); s.set_lap_wrap(True)
ts = []
for i in range(300):
    t0 = time.perf_counter(); u0, st = s.compute_control(x0, 40.0); ts.append(time.perf_counter() - t0)
print("compute_control p50 %.3f ms p99 %.3f ms" % (np.percentile(ts[50:], 50) * 1e3, np.percentile(ts[50:], 99) * 1e3))
