"""Measures the BASELINE.json configurations that fit one GPU (beyond the bench.py headline run).

    python tools/measure_configs.py            # on the GPU box; prints one JSON line per configuration
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from bench import closed_loop_config5, rti_throughput  # noqa: E402,F401  (the workloads live in bench.py; this tool runs the whole list)


if __name__ == "__main__":
    all_tracks = bench.ALL_TRACKS      # all seven tracks of data/, as tests/test_gpu_configs.py
    for fn, kw in ((rti_throughput, dict(model="fkin6", B=1024)), (rti_throughput, dict(model="fkin6", B=1024, host_state=True)),
                   (rti_throughput, dict(model="fkin6", B=8192)),
                   (rti_throughput, dict(model="fkin6", B=1024, sqp="FIXED_STEP")),
                   (rti_throughput, dict(model="fkin6", B=1024, sqp="MERIT_BACKTRACKING")),
                   (rti_throughput, dict(model="fkin6", B=1024, persistent=True, steps=500, warmup=20)),
                   (rti_throughput, dict(model="fkin6", B=1024, sqp="FIXED_STEP", persistent=True, steps=500, warmup=20)),
                   (rti_throughput, dict(model="fkin6", B=1024, sqp="MERIT_BACKTRACKING", persistent=True, steps=500, warmup=20)),
                   (rti_throughput, dict(model="fkin6", B=1024, track_rows="soft")),
                   (rti_throughput, dict(model="fkin6", B=1024, track_rows="soft", persistent=True, steps=500, warmup=20)),
                   (rti_throughput, dict(model="fdyn6", B=8192, terminal_bounds="stage")),
                   (rti_throughput, dict(model="fdyn6", B=8192, terminal_bounds="stage", track_rows="soft")),
                   (rti_throughput, dict(model="fdyn6", B=8192, track_rows="soft")),
                   (rti_throughput, dict(model="fdyn6", B=8192, tracks=all_tracks, terminal_bounds="stage")),
                   (closed_loop_config5, dict(B=4096, steps=200)),
                   (closed_loop_config5, dict(B=4096, steps=200, terminal_bounds="stage")),
                   # the same configurations with the un-crossed dynamic model (named deviation from quirk Q3)
                   (rti_throughput, dict(model="fdyn6u", B=8192, track_rows="soft")),
                   (rti_throughput, dict(model="fdyn6u", B=8192, terminal_bounds="stage", track_rows="soft", recover=True)),
                   (rti_throughput, dict(model="fdyn6u", B=8192, tracks=all_tracks, terminal_bounds="stage", track_rows="soft", recover=True)),
                   # the reference's live integrator (python/main.py:234-236): IRK, 4 Gauss-Legendre stages, one step per interval
                   (rti_throughput, dict(model="fkin6", B=1024, integrator="IRK")),
                   (rti_throughput, dict(model="fkin6", B=8192, integrator="IRK")),
                   (rti_throughput, dict(model="fdyn6u", B=8192, terminal_bounds="stage", track_rows="soft", recover=True, integrator="IRK")),
                   # the live options as a whole (python/main.py:227-238): SQP x 2, MERIT_BACKTRACKING, IRK
                   (rti_throughput, dict(model="fkin6", B=1024, sqp="MERIT_BACKTRACKING", integrator="IRK")),
                   (rti_throughput, dict(model="fkin6", B=8192, sqp="MERIT_BACKTRACKING", integrator="IRK")),
                   # the same in the persistent loop (IRK inside k_steps)
                   (rti_throughput, dict(model="fkin6", B=1024, integrator="IRK", persistent=True, steps=500, warmup=20)),
                   (rti_throughput, dict(model="fkin6", B=1024, sqp="MERIT_BACKTRACKING", integrator="IRK", persistent=True, steps=500, warmup=20)),
                   (rti_throughput, dict(model="fkin6", B=1024, sqp="FIXED_STEP", integrator="IRK", persistent=True, steps=500, warmup=20)),
                   # round 3: the persistent loop on the soft track-row tables with the collocation integrator, and with the plants by Radau IIA --
                   # the live options (python/main.py:227-238) with the live plant integrator (:395-400) in one launch
                   (rti_throughput, dict(model="fkin6", B=1024, track_rows="soft", integrator="IRK", persistent=True, steps=200, warmup=20)),
                   (rti_throughput, dict(model="fkin6", B=1024, plant_integrator="IRK", persistent=True, steps=200, warmup=20)),
                   (rti_throughput, dict(model="fkin6", B=1024, sqp="MERIT_BACKTRACKING", integrator="IRK", plant_integrator="IRK", persistent=True, steps=200, warmup=20)),
                   (rti_throughput, dict(model="fkin6", B=1024, sqp="MERIT_BACKTRACKING", integrator="IRK", plant_integrator="IRK", persistent=False, steps=20, warmup=5)),
                   (closed_loop_config5, dict(B=4096, steps=200, terminal_bounds="stage", plant="KIN6_DYN6U")),
                   (closed_loop_config5, dict(B=4096, steps=200, terminal_bounds="stage", plant="KIN6_DYN6U", soft_state_bounds=(1000.0, 1000.0))),
                   (closed_loop_config5, dict(B=4096, steps=200, device_loop=True)),
                   (closed_loop_config5, dict(B=4096, steps=200, terminal_bounds="stage", plant="KIN6_DYN6U", soft_state_bounds=(1000.0, 1000.0), device_loop=True)),
                   # the per-GPU share of configs[4] on 8 GPUs: 512 cars
                   (closed_loop_config5, dict(B=512, steps=200, terminal_bounds="stage", plant="KIN6_DYN6U", soft_state_bounds=(1000.0, 1000.0), device_loop=True)),
                   (closed_loop_config5, dict(B=512, steps=200, terminal_bounds="stage", plant="KIN6_DYN6U", soft_state_bounds=(1000.0, 1000.0), persistent=True))):
        print(json.dumps(fn(**kw)), flush=True)
