"""configs[2] (dynamic bicycle + soft track rows): the one-launch persistent loop against launches per step, at the resident batch (1024) and at
B = 8192 (IHM2MPC_PERSISTENT_ROUNDS=1: workgroups in rounds of whole histories).  VERDICT r3 weak 7 / next 2.

    python tools/probes/dyn_persistent_vs_perstep.py            # on the GPU box
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

rows = []
for B in (1024, 8192):
    for integ in ("ERK", "IRK"):
        for persistent in (False, True):
            if persistent and B > 1024:
                os.environ["IHM2MPC_PERSISTENT_ROUNDS"] = "1"
            r = bench.rti_throughput(model="fdyn6u", B=B, steps=20, warmup=5, terminal_bounds="stage", track_rows="soft", recover=not persistent, integrator=integ,
                                     persistent=persistent)
            rows.append(dict(B=B, integrator=integ, persistent=persistent, solves_per_s=r["solves_per_s"], ms_per_step=r["ms_per_step"], ok_fraction=r["ok_fraction"]))
            print(json.dumps(rows[-1]), flush=True)
