import sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
from measure_configs import rti_throughput
for kw in (dict(model="fkin6", B=1024, sqp="MERIT_BACKTRACKING", integrator="IRK"), dict(model="fkin6", B=8192, sqp="MERIT_BACKTRACKING", integrator="IRK"),
           dict(model="fkin6", B=1024, sqp="MERIT_BACKTRACKING"), dict(model="fkin6", B=8192, sqp="MERIT_BACKTRACKING")):
    r = rti_throughput(**kw); print(kw.get("integrator", "ERK"), kw["B"], round(r["solves_per_s"]), round(r["ms_per_step"], 3), r["status"], round(r["linearize_ms"],3), round(r["qp_ms"],3), r.get("alpha_lt1"))
