"""configs[2] (fdyn6u + soft track rows, B = 8192): what the re-initialisation of failed instances between steps costs per step.
    python tools/probes/config2_recover_cost.py            # on the GPU box
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
for integ in ("ERK", "IRK"):
    for rec in (True, False):
        r = bench.rti_throughput(model="fdyn6u", B=8192, steps=20, warmup=5, terminal_bounds="stage", track_rows="soft", recover=rec, integrator=integ)
        print(json.dumps(dict(integrator=integ, recover=rec, solves_per_s=r["solves_per_s"], ms_per_step=r["ms_per_step"], linearize_ms=r["linearize_ms"], qp_ms=r["qp_ms"],
                              ok_fraction=r["ok_fraction"])), flush=True)
