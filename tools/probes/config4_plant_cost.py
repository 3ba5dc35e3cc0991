"""configs[4] share of one GPU (512 cars x 200 periods in one persistent launch): what the plant costs per control period -- the plant's sub-steps
per period varied, and the kinematic plant (which rides on the linearisation's spare lane) for comparison.
    python tools/probes/config4_plant_cost.py            # on the GPU box
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ihm2_amd.closed_loop_sim import SimModelVariant, Simulator, SimulatorConfig, run_closed_loop_persistent  # noqa: E402
from ihm2_amd.controller import IHM2Controller  # noqa: E402
from ihm2_amd.track import track_table  # noqa: E402

plan = track_table(bench.TRACK)
B, steps = 512, 200
x0 = bench.sample_x0(plan, B, seed=5)
for plant, nsub in (("KIN6_DYN6U", 100), ("KIN6_DYN6U", 50), ("KIN6_DYN6U", 25), ("KIN6", 25), ("DYN6U", 100)):
    ctrl = IHM2Controller(plan.s_ref, plan.kappa_ref, batch_size=B, terminal_bounds="stage", soft_state_bounds=(1000.0, 1000.0))
    sim = Simulator(ctrl, SimulatorConfig(sampling_time=ctrl.dt, num_steps=nsub), SimModelVariant[plant])
    ctrl.warm_start(x0)
    t0 = time.perf_counter()
    res = run_closed_loop_persistent(ctrl, sim, x0, steps, lap_length=plan.lap_length)
    el = time.perf_counter() - t0
    print(json.dumps(dict(plant=plant, plant_substeps=nsub, wall_s=el, ms_per_period=el / res.u.shape[0] * 1e3, control_steps_per_s=B * res.u.shape[0] / el,
                          alive=int(res.alive.sum()), finished=int(res.finished.sum()))), flush=True)
    ctrl.solver.free()
