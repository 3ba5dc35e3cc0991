"""Persistent loop on soft tables: B = 1024 kinematic cars with soft track rows (k_steps<8,3,1,1,0,0,0>), and configs[4]'s per-GPU share
(k_steps<8,2,0,1,...>, soft plant-state bounds).
    python tools/probes/soft_persistent_cost.py            # on the GPU box
"""
import json, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
if len(sys.argv) > 1:       # a diagnostic build of the library
    from ihm2_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
for steps in (20, 200, 20, 200):
    r = bench.rti_throughput(model="fkin6", B=1024, steps=steps, warmup=5, track_rows="soft", persistent=True)
    print(json.dumps(dict(case="fkin6 + soft track rows, persistent", steps=steps, solves_per_s=r["solves_per_s"], ms_per_step=r["ms_per_step"], ok_fraction=r["ok_fraction"])), flush=True)
