"""What the fifteenth row (a_lat, `k_qp_wave<10,4,2,1>`) costs: B = 1024 kinematic cars with soft track rows, launches per step, with and
without the row.
    python tools/probes/alat_row_cost.py            # on the GPU box
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
for rows in ("soft", "soft+a_lat"):
    for B in (1024, 8192):
        r = bench.rti_throughput(model="fkin6", B=B, steps=20, warmup=5, track_rows=rows, persistent=False)
        print(json.dumps(dict(track_rows=rows, B=B, solves_per_s=r["solves_per_s"], ms_per_step=r["ms_per_step"], linearize_ms=r["linearize_ms"], qp_ms=r["qp_ms"],
                              ok_fraction=r["ok_fraction"], mean_qp_iter=r.get("mean_qp_iter"))), flush=True)
