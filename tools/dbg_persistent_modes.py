import sys, json
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
from measure_configs import rti_throughput
for kw in (dict(model="fkin6", B=1024, persistent=True, steps=500, warmup=20),
           dict(model="fkin6", B=1024, sqp="FIXED_STEP", persistent=True, steps=500, warmup=20),
           dict(model="fkin6", B=1024, sqp="MERIT_BACKTRACKING", persistent=True, steps=500, warmup=20),
           dict(model="fkin6", B=1024, track_rows="soft", persistent=True, steps=500, warmup=20)):
    r = rti_throughput(**kw); print(kw.get("sqp"), kw.get("track_rows"), round(r["solves_per_s"]), round(r["ms_per_step"], 3))
