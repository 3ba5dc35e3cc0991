"""Diagnostic: the persistent loop (500 steps, B = 1024) in its four modes -- RTI, SQP FIXED_STEP, SQP MERIT_BACKTRACKING, RTI with soft
track rows -- solves/s and ms per step (a quick regression check of the register-starved k_steps instantiations after a kernel change)."""
import sys, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
from measure_configs import rti_throughput
for kw in (dict(model="fkin6", B=1024, persistent=True, steps=500, warmup=20),
           dict(model="fkin6", B=1024, sqp="FIXED_STEP", persistent=True, steps=500, warmup=20),
           dict(model="fkin6", B=1024, sqp="MERIT_BACKTRACKING", persistent=True, steps=500, warmup=20),
           dict(model="fkin6", B=1024, track_rows="soft", persistent=True, steps=500, warmup=20)):
    r = rti_throughput(**kw); print(kw.get("sqp"), kw.get("track_rows"), round(r["solves_per_s"]), round(r["ms_per_step"], 3))
