"""Persistent loop with the collocation integrator on the shooting intervals (RTI and the live SQP options), B = 1024:

    python3 tools/run_persistent_irk.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import measure_configs as m  # noqa: E402

if __name__ == "__main__":
    for kw in (dict(integrator="IRK"), dict(integrator="IRK", persistent=True, steps=500, warmup=20), dict(integrator="ERK", persistent=True, steps=500, warmup=20),
               dict(integrator="IRK", sqp="MERIT_BACKTRACKING"), dict(integrator="IRK", sqp="MERIT_BACKTRACKING", persistent=True, steps=500, warmup=20),
               dict(integrator="IRK", sqp="FIXED_STEP", persistent=True, steps=500, warmup=20)):
        print(json.dumps(m.rti_throughput(model="fkin6", B=1024, **kw)), flush=True)
