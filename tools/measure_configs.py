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
from ihm2_amd import ocp as O  # noqa: E402
from ihm2_amd.closed_loop_sim import SimModelVariant, Simulator, SimulatorConfig, run_closed_loop, run_closed_loop_device, run_closed_loop_persistent  # noqa: E402
from ihm2_amd.controller import IHM2Controller  # noqa: E402
from ihm2_amd.solver import BatchedOcpSolver  # noqa: E402
from ihm2_amd.track import track_table  # noqa: E402


def rti_throughput(model, B, steps=20, warmup=5, tracks=("fsds_competition_1",), terminal_bounds="reference", track_rows=None, recover=False, host_state=False,
                   sqp=None, persistent=False, integrator="ERK"):
    plans = [track_table(t) for t in tracks]
    mdl = O.get_acados_model_from_explicit_dynamics("ihm2_" + model, {"fkin6": O.fkin6_model, "fdyn6": O.fdyn6_model, "fdyn6u": O.fdyn6u_model}[model], 8, 2, 3000)
    ocp = O.get_acados_ocp(mdl, 40, 2.0, 31.0, 500.0, 0.5, 1e6, 1.0)
    ocp.cost.W, ocp.cost.W_e = O.default_weights()
    ocp.solver_options.tf = 2.0
    if integrator == "IRK":     # python/main.py:234-236: IRK, 4 Gauss-Legendre stages, one step per shooting interval
        ocp.solver_options.integrator_type, ocp.solver_options.sim_method_num_steps = "IRK", 1
    if sqp is not None:         # the live options of python/main.py:230-237: "SQP", max_iter 2, "MERIT_BACKTRACKING"
        ocp.solver_options.nlp_solver_type, ocp.solver_options.nlp_solver_max_iter, ocp.solver_options.globalization = "SQP", 2, sqp
    if terminal_bounds == "stage":      # see IHM2Controller(terminal_bounds=...): quirk Q1
        c = ocp.constraints
        c.idxbx_e, c.lbx_e, c.ubx_e = c.idxbx.copy(), c.lbx.copy(), c.ubx.copy()
    widths = None
    if track_rows is not None:          # nonlinear track-boundary rows, soft 100/100 as old/generate_acaods_interface.py:380-449
        ocp.model.con_h_expr = "track"
        c = ocp.constraints
        c.lh = c.lh_e = np.array([-1e3, -1e3]); c.uh = c.uh_e = np.array([0.0, 0.0])
        if track_rows == "soft":
            c.idxsh, c.idxsh_e = np.arange(2), np.arange(2)
            ocp.cost.zl = ocp.cost.zu = ocp.cost.Zl = ocp.cost.Zu = np.full(2, 100.0)
            ocp.cost.zl_e = ocp.cost.zu_e = ocp.cost.Zl_e = ocp.cost.Zu_e = np.full(2, 100.0)
        widths = np.array([[p.right_widths.min(), p.left_widths.min()] for p in plans])
    s_ref = np.stack([p.s_ref for p in plans]); k_ref = np.stack([p.kappa_ref for p in plans])
    tid = (np.arange(B) % len(plans)).astype(np.int32)
    solver = BatchedOcpSolver(ocp, B, s_ref, k_ref, track_id=tid, track_widths=widths)
    x0 = np.zeros((B, 8))
    for t, p in enumerate(plans):
        sel = tid == t
        x0[sel] = bench.sample_x0(p, int(sel.sum()), seed=20240607 + t)
    solver.set_x0(x0); solver.init_guess()
    solver.set_lap_wrap(True)        # long runs: cars that complete a lap are moved back by one lap length (the tables hold three laps)
    if model != "fkin6":          # a few SQP iterations on the frozen problem: the kinematic rollout is not a dynamic trajectory
        yref = np.zeros((B, 40, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(40)[None] / 40
        yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
        solver.set_yref(yref); solver.set_yref_e(yref_e); solver.solve(3)
    plant = {"fkin6": 0, "fdyn6": 1, "fdyn6u": 2}[model]

    def step():
        if host_state:      # the state crosses the boundary as host buffers every step: D2H (B,8) + H2D (B,8) on top of the u0 readback
            solver.sim_advance(model=plant, M_sim=25)
            solver.set_x0(solver.get_x0())
            solver.prepare_step(40.0); solver.solve_async()
        else:
            solver.step(40.0, model=plant, M_sim=25)
        u0 = solver.get_u0()
        if recover:
            solver.reinit_failed()
        return u0

    for _ in range(warmup):
        step()
    solver.synchronize(); t0 = time.perf_counter(); tl = tq = 0.0; n_ok = 0
    if persistent:      # all steps in one launch (ihm2mpc_run_steps), histories read back at the end
        solver.reserve_history(steps)
        solver.synchronize(); t0 = time.perf_counter()
        h = solver.run_steps(40.0, steps, model=plant, M_sim=25, u0_hist=True, status_hist=True)
        n_ok = int(np.isin(h["status"], (0, 2) if sqp else (0,)).sum())
    else:
        for _ in range(steps):
            step(); tm = solver.get_timings(); tl += tm["linearize_ms"]; tq += tm["qp_ms"]; n_ok += int(np.isin(solver.get_status(), (0, 2) if sqp else (0,)).sum())
    solver.synchronize(); el = time.perf_counter() - t0
    st = solver.get_status()
    out = dict(model=model, B=B, tracks=len(plans), terminal_bounds=terminal_bounds, track_rows=track_rows, recover=recover, host_state=host_state, sqp=sqp,
               persistent=persistent, integrator=integrator, steps=steps, solves_per_s=B * steps / el, ms_per_step=el / steps * 1e3, linearize_ms=tl / steps,
               qp_ms=tq / steps, ok_fraction=n_ok / (B * steps), status={str(k): int(v) for k, v in enumerate(np.bincount(st, minlength=5)) if v},
               qp_iter_mean=float(solver.get_qp_iter().mean()))
    if sqp:
        ss = solver.get_sqp_stats()
        out["sqp_iter_mean"] = float(ss["sqp_iter"].mean()); out["alpha_mean"] = float(ss["alpha"].mean()); out["alpha_lt1"] = float((ss["alpha"] < 1.0).mean())
    solver.free()
    return out


def closed_loop_config5(B=4096, steps=200, terminal_bounds="reference", plant="KIN6_DYN6", soft_state_bounds=None, device_loop=False, persistent=False):
    plan = track_table("fsds_competition_1")
    ctrl = IHM2Controller(plan.s_ref, plan.kappa_ref, batch_size=B, terminal_bounds=terminal_bounds, soft_state_bounds=soft_state_bounds)
    sim = Simulator(ctrl, SimulatorConfig(sampling_time=ctrl.dt, num_steps=100), SimModelVariant[plant])
    x0 = bench.sample_x0(plan, B, seed=5)
    ctrl.warm_start(x0)
    t0 = time.perf_counter()
    runner = run_closed_loop_persistent if persistent else (run_closed_loop_device if device_loop else run_closed_loop)
    res = runner(ctrl, sim, x0, steps, lap_length=plan.lap_length)
    el = time.perf_counter() - t0
    st = res.stats()
    return dict(config=5, terminal_bounds=terminal_bounds, plant=plant, soft_state_bounds=soft_state_bounds, device_loop=device_loop, persistent=persistent, B=B, steps=int(res.u.shape[0]), wall_s=el, control_steps_per_s=B * res.u.shape[0] / el, alive=int(res.alive.sum()),
                finished=int(res.finished.sum()), failed=st["failed"], mean_speed=st["mean_speed"],
                progress_m_median=float(np.median(res.x[-1, :, 0] - res.x[0, :, 0])))


if __name__ == "__main__":
    all_tracks = ("fsds_competition_1", "fsds_competition_2", "fsds_competition_3", "fsds_default")
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
                   (closed_loop_config5, dict(B=4096, steps=200, terminal_bounds="stage", plant="KIN6_DYN6U")),
                   (closed_loop_config5, dict(B=4096, steps=200, terminal_bounds="stage", plant="KIN6_DYN6U", soft_state_bounds=(1000.0, 1000.0))),
                   (closed_loop_config5, dict(B=4096, steps=200, device_loop=True)),
                   (closed_loop_config5, dict(B=4096, steps=200, terminal_bounds="stage", plant="KIN6_DYN6U", soft_state_bounds=(1000.0, 1000.0), device_loop=True)),
                   # the per-GPU share of configs[4] on 8 GPUs: 512 cars
                   (closed_loop_config5, dict(B=512, steps=200, terminal_bounds="stage", plant="KIN6_DYN6U", soft_state_bounds=(1000.0, 1000.0), device_loop=True)),
                   (closed_loop_config5, dict(B=512, steps=200, terminal_bounds="stage", plant="KIN6_DYN6U", soft_state_bounds=(1000.0, 1000.0), persistent=True))):
        print(json.dumps(fn(**kw)), flush=True)
