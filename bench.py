#!/usr/bin/env python3
"""bench.py -- NMPC RTI solves/s of the batched bicycle NMPC on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W                 # N > 1: bench.py starts the N ranks itself (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      # or a launcher does

Default workload (BASELINE.json configs[1], per GPU): batch = 1024 kinematic bicycle (fkin6) instances, N = 40,
dt = 0.05, RK4 x M = 25, track fsds_competition_1, synthetic Monte-Carlo initial states of SURVEY.md 8d.
One "step" = one closed-loop control step of the whole batch, everything resident on the device:
plant advance (kinematic model, RK4 x 25) -> reference ramp + warm-start shift (python/main.py:303-322)
-> one SQP-RTI iteration (linearise, QP, full step) -> read back u0 (B x 2) to the host.
Weak scaling: every rank owns its own instances on its own GPU; no data-path collective; RCCL is used for the
barrier / max-reduce of the timing and for one final all_gather of the results.

    --config 2   BASELINE.json configs[2]: batch 8192 dynamic bicycle + soft nonlinear track rows (one GPU)
    --config 3   configs[3]: 65 536 dynamic bicycles over all seven data/ tracks, sharded (per-GPU share 8192; N = 8 is the whole of it)
    --config 4   configs[4]: the MiL loop, 4096 cars x 200 control steps, sharded (per-GPU share 512)

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (live HIP-event time on the solver's stream);
`cpu_baseline` is the CPU oracle (own restatement, NOT acados) on the host cores; `configs2` / `live_options` are short runs of
BASELINE.json configs[2] and of the reference's live solver options, measured after the timed region of the headline (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_H, DT, M_SUB, S_TARGET = 40, 0.05, 25, 40.0
TRACK = "fsds_competition_1"
ALL_TRACKS = ("acceleration", "fsds_competition_1", "fsds_competition_2", "fsds_competition_3", "fsds_default", "short_skidpad", "skidpad")
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector peak = fp64 matrix peak (guide: MI355X_MICROARCH.md / SURVEY.md 8d)
HBM_PEAK_GBS = 8000.0
CONFIG3_TOTAL, CONFIG4_TOTAL, NODE_GPUS = 65536, 4096, 8
TRAFFIC_JSON = os.path.join("profiles", "r4", "bench_traffic.json")


def build_problem(batch):
    from ihm2_amd import ocp as O
    from ihm2_amd.track import track_table

    track = track_table(TRACK)
    model = O.get_acados_model_from_explicit_dynamics("ihm2_fkin6", O.fkin6_model, 8, 2, 3000)
    ocp = O.get_acados_ocp(model, N_H, 2.0, 31.0, 500.0, 0.5, 1e6, 1.0)      # python/main.py:186-192 defaults
    ocp.cost.W, ocp.cost.W_e = O.default_weights()                            # python/main.py:193-210
    ocp.solver_options.tf = N_H * DT
    ocp.solver_options.sim_method_num_steps = M_SUB
    return ocp, track


def sample_x0(track, B, seed):
    """SURVEY.md section 8d."""
    from ihm2_amd.constants import l_R

    rng = np.random.default_rng(seed)
    s = rng.uniform(0, track.lap_length, B)
    n = rng.uniform(-0.5, 0.5, B)
    psi = rng.uniform(-0.1, 0.1, B)
    vx = rng.uniform(2, 15, B)
    kap = np.interp(s, track.s_ref, track.kappa_ref)
    T = rng.uniform(-100, 300, B)
    delta = np.arctan(2 * np.tan(np.arcsin(np.clip(kap * l_R, -0.9, 0.9))))
    return np.stack([s, n, psi, vx, 0 * s, vx * kap, T, delta], 1)


def monte_carlo_batch(plans, total, seed=20240607):
    """ONE global Monte-Carlo batch over several tracks (BASELINE.json configs[3]): instance g drives on track g mod len(plans), its
    initial state is the next draw of that track's own stream.  Returns (x0 (total, 8), track_id (total,)); a rank takes the rows
    [lo, hi) of both (ihm2_amd/dist.py::shard_bounds), so the concatenation over the ranks IS the single-handle batch."""
    tid = (np.arange(total) % len(plans)).astype(np.int32)
    x0 = np.zeros((total, 8))
    for t, p in enumerate(plans):
        sel = tid == t
        x0[sel] = sample_x0(p, int(sel.sum()), seed=seed + t)
    return x0, tid


def flops_per_solve(n_ipm):
    """Algorithmic flop model of SURVEY.md 8d (fkin6): linearisation + structure-exploiting QP."""
    nx, nu = 8, 2
    lin = N_H * M_SUB * 4 * (2 * nx * nx * (nx + nu) + 400)
    qp = n_ipm * N_H * 1800.0
    return lin, qp


# --------------------------------------------------------------------------------------------------------------- workloads beyond the headline
def rti_throughput(model, B, steps=20, warmup=5, tracks=(TRACK,), terminal_bounds="reference", track_rows=None, recover=False, host_state=False,
                   sqp=None, persistent=False, integrator="ERK", device=0, x0=None, track_id=None, ranks=None, plant_integrator="ERK"):
    """Closed-loop RTI throughput of one batch on one device (the workloads of BASELINE.json configs[1..3] and their variants).
    ``x0`` / ``track_id``: this rank's rows of a global batch (default: the per-track draws of :func:`monte_carlo_batch`).
    ``ranks``: a RankContext -- the timed region is then bracketed by its barrier."""
    from ihm2_amd import ocp as O
    from ihm2_amd.solver import BatchedOcpSolver
    from ihm2_amd.track import track_table

    plans = [track_table(t) for t in tracks]
    mdl = O.get_acados_model_from_explicit_dynamics("ihm2_" + model, {"fkin6": O.fkin6_model, "fdyn6": O.fdyn6_model, "fdyn6u": O.fdyn6u_model}[model], 8, 2, 3000)
    ocp = O.get_acados_ocp(mdl, 40, 2.0, 31.0, 500.0, 0.5, 1e6, 1.0)
    ocp.cost.W, ocp.cost.W_e = O.default_weights()
    ocp.solver_options.tf = 2.0
    if integrator == "IRK":     # python/main.py:234-236: IRK, 4 Gauss-Legendre stages, one step per shooting interval
        ocp.solver_options.integrator_type, ocp.solver_options.sim_method_num_steps = "IRK", 1
    if plant_integrator == "IRK":     # python/main.py:395-400: the plants by IRK, GAUSS_RADAU_IIA, 4 stages (x M_sim steps)
        ocp.solver_options.sim_integrator_type, ocp.solver_options.sim_collocation_type = "IRK", "GAUSS_RADAU_IIA"
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
        nh = 2
        if track_rows == "soft+a_lat":  # the kinematic set with its fifth row, |a_lat| <= 5 m/s^2 (:198-209, :52-53), at the stages only
            ocp.model.con_h_expr, nh = "track+a_lat", 3
            c.lh = np.array([-1e3, -1e3, -5.0]); c.uh = np.array([0.0, 0.0, 5.0])
        if track_rows in ("soft", "soft+a_lat"):
            c.idxsh, c.idxsh_e = np.arange(nh), np.arange(2)
            ocp.cost.zl = ocp.cost.zu = ocp.cost.Zl = ocp.cost.Zu = np.full(nh, 100.0)
            ocp.cost.zl_e = ocp.cost.zu_e = ocp.cost.Zl_e = ocp.cost.Zu_e = np.full(2, 100.0)
        widths = np.array([[p.right_widths.min(), p.left_widths.min()] for p in plans])
    s_ref = np.stack([p.s_ref for p in plans]); k_ref = np.stack([p.kappa_ref for p in plans])
    if x0 is None:
        x0, track_id = monte_carlo_batch(plans, B)
    tid = np.ascontiguousarray(track_id, dtype=np.int32)
    solver = BatchedOcpSolver(ocp, B, s_ref, k_ref, track_id=tid, track_widths=widths, device=device)
    solver.set_x0(x0); solver.init_guess()
    solver.set_lap_wrap(True)        # long runs: cars that complete a lap are moved back by one lap length (the tables hold three laps)
    if model != "fkin6":          # a few SQP iterations on the frozen problem: the kinematic rollout is not a dynamic trajectory
        yref = np.zeros((B, 40, 12)); yref[:, :, 0] = x0[:, 0:1] + 40.0 * np.arange(40)[None] / 40
        yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 40.0
        solver.set_yref(yref); solver.set_yref_e(yref_e); solver.solve(3)
    plant = {"fkin6": 0, "fdyn6": 1, "fdyn6u": 2}[model]
    if ranks is not None:
        ranks.attach(solver, ranks.total)

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
    tl = tq = 0.0; n_ok = 0
    good = (0, 2) if sqp else (0,)
    if persistent:
        solver.reserve_history(steps)
    if ranks is not None:
        ranks.barrier(solver)
    solver.synchronize(); t0 = time.perf_counter()
    if persistent:      # all steps in one launch (ihm2mpc_run_steps), histories read back at the end
        h = solver.run_steps(40.0, steps, model=plant, M_sim=25, u0_hist=True, status_hist=True)
        n_ok = int(np.isin(h["status"], good).sum())
    else:
        for _ in range(steps):
            step(); tm = solver.get_timings(); tl += tm["linearize_ms"]; tq += tm["qp_ms"]; n_ok += int(np.isin(solver.get_status(), good).sum())
    if ranks is not None:
        ranks.barrier(solver)
    solver.synchronize(); el = time.perf_counter() - t0
    st = solver.get_status()
    out = dict(model=model, B=B, tracks=len(plans), terminal_bounds=terminal_bounds, track_rows=track_rows, recover=recover, host_state=host_state, sqp=sqp,
               persistent=persistent, integrator=integrator, plant_integrator=plant_integrator, steps=steps, elapsed_s=el, solves_per_s=B * steps / el, ms_per_step=el / steps * 1e3, linearize_ms=tl / steps,
               qp_ms=tq / steps, ok_fraction=n_ok / (B * steps), n_ok=n_ok, status={str(k): int(v) for k, v in enumerate(np.bincount(st, minlength=5)) if v},
               qp_iter_mean=float(solver.get_qp_iter().mean()))
    if sqp:
        ss = solver.get_sqp_stats()
        out["sqp_iter_mean"] = float(ss["sqp_iter"].mean()); out["alpha_mean"] = float(ss["alpha"].mean()); out["alpha_lt1"] = float((ss["alpha"] < 1.0).mean())
    if ranks is not None:
        # everything that needs this rank's communicator happens BEFORE the solver goes: ihm2mpc_free takes the handle's communicator with it
        tg0 = time.perf_counter()
        out["gathered"] = ranks.gather_results(solver)
        out["gather_ms"] = (time.perf_counter() - tg0) * 1e3 if ranks.carrier != "none" else 0.0
        out["elapsed_max_s"] = ranks.max(el)
        out["n_ok_sum"] = ranks.sum(n_ok)
        out["collective"] = ranks.verify(solver)
        ranks.detach()
    solver.free()
    return out


def closed_loop_config5(B=4096, steps=200, terminal_bounds="reference", plant="KIN6_DYN6", soft_state_bounds=None, device_loop=False, persistent=False,
                        device=0, x0=None, ranks=None, plant_integrator="ERK"):
    """The MiL loop of BASELINE.json configs[4] (new_python/closed_loop_sim.py:34-42, python/main.py:448-517) for B cars on one device."""
    from ihm2_amd.closed_loop_sim import SimModelVariant, Simulator, SimulatorConfig, run_closed_loop, run_closed_loop_device, run_closed_loop_persistent
    from ihm2_amd.controller import IHM2Controller
    from ihm2_amd.track import track_table

    plan = track_table(TRACK)
    # plant_integrator "IRK": the plants as the reference configures them (python/main.py:395-400: IRK, GAUSS_RADAU_IIA, 4 stages, 100 steps per period)
    ctrl = IHM2Controller(plan.s_ref, plan.kappa_ref, batch_size=B, terminal_bounds=terminal_bounds, soft_state_bounds=soft_state_bounds, device=device,
                          sim_integrator_type=plant_integrator)
    sim = Simulator(ctrl, SimulatorConfig(sampling_time=ctrl.dt, num_steps=100), SimModelVariant[plant])
    if x0 is None:
        x0 = sample_x0(plan, B, seed=5)
    ctrl.warm_start(x0)
    if ranks is not None:
        ranks.attach(ctrl.solver, ranks.total)
        ranks.barrier(ctrl.solver)
    t0 = time.perf_counter()
    runner = run_closed_loop_persistent if persistent else (run_closed_loop_device if device_loop else run_closed_loop)
    res = runner(ctrl, sim, x0, steps, lap_length=plan.lap_length)
    if ranks is not None:
        ranks.barrier(ctrl.solver)
    el = time.perf_counter() - t0
    st = res.stats()
    out = dict(config=5, terminal_bounds=terminal_bounds, plant=plant, plant_integrator=plant_integrator, soft_state_bounds=soft_state_bounds, device_loop=device_loop, persistent=persistent, B=B,
               steps=int(res.u.shape[0]), wall_s=el, control_steps_per_s=B * res.u.shape[0] / el, alive=int(res.alive.sum()),
               finished=int(res.finished.sum()), failed=st["failed"], mean_speed=st["mean_speed"],
               progress_m_median=float(np.median(res.x[-1, :, 0] - res.x[0, :, 0])))
    if ranks is not None:
        tg0 = time.perf_counter()
        out["gathered"] = ranks.gather_results(ctrl.solver)
        out["gather_ms"] = (time.perf_counter() - tg0) * 1e3 if ranks.carrier != "none" else 0.0
        out["elapsed_max_s"] = ranks.max(el)
        for k in ("alive", "finished", "failed"):
            out[k + "_sum"] = int(ranks.sum(out[k]))
        out["collective"] = ranks.verify(ctrl.solver)
        ranks.detach()
    return out


# --------------------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(ocp, track, x0_all, budget_s=12.0, gpu_state=None, device=0):
    """Oracle (own C restatement, OpenMP over instances) on a bounded sample of the same workload; with gpu_state (the iterate,
    multipliers and plant state of a sample of instances AFTER the timed region) also the same-run check of SURVEY.md 8d: one
    more control step of that sample on the GPU and in the oracle, deviation of the two results."""
    from oracle import oracle as orc

    P = orc.OracleProblem(ocp.flatten().as_dict(track.s_ref, track.kappa_ref))
    check = None
    if gpu_state is not None:
        from ihm2_amd.solver import BatchedOcpSolver

        xs, us, x0s, pis, lams = (np.ascontiguousarray(a) for a in gpu_state)
        Bs = xs.shape[0]
        small = BatchedOcpSolver(ocp, Bs, track.s_ref, track.kappa_ref, device=device)
        small.set_x(xs); small.set_u(us); small.set_x0(x0s); small.set_multipliers(pis, lams)
        small.prepare_step(S_TARGET)
        st_g = small.solve()
        xo, uo = xs.copy(), us.copy()
        yref, yref_e = orc.prepare_step(N_H, x0s, S_TARGET, xo, uo)
        out = P.rti_step(xo, uo, x0s, yref, yref_e, pi=pis, lam=lams)
        ok = (st_g == 0) & (out["status"] == 0)
        rel = lambda a, b: float(np.max(np.abs(a - b) / (1.0 + np.abs(b)))) if a.size else 0.0
        check = {"instances": int(Bs), "both_status_0": int(ok.sum()), "status_equal": bool(np.array_equal(st_g, out["status"])),
                 "qp_iter_equal": bool(np.array_equal(small.get_qp_iter()[ok], out["qp_iter"][ok])),
                 "max_rel_dev_x": rel(small.get_x()[ok], xo[ok]), "max_rel_dev_u": rel(small.get_u()[ok], uo[ok]),
                 "max_rel_dev_kkt_residuals": rel(small.get_residuals(), out["res"])}
        small.free()
    # the GPU box grants this job a 16-core CPU share per GPU; IHM2_CPU_THREADS overrides
    threads = int(os.environ.get("IHM2_CPU_THREADS", min(orc.num_threads(), len(os.sched_getaffinity(0)), 16)))
    Bs = min(512, x0_all.shape[0])
    x0 = x0_all[:Bs].copy()
    x, u = orc.stanley_guess(P, track.s_ref, track.kappa_ref, x0, N_H, M_SUB)      # the warm start the GPU run takes (ihm2mpc_init_guess)
    pi = lam = None
    xcur = x0
    solves, t_acc, steps = 0, 0.0, 0
    while t_acc < budget_s and steps < 200:
        xcur = P.sim_step(xcur, u[:, 0].copy(), 0, M_SUB, nthreads=threads)
        t0 = time.perf_counter()
        yref, yref_e = orc.prepare_step(N_H, xcur, S_TARGET, x, u)
        out = P.rti_step(x, u, xcur, yref, yref_e, pi=pi, lam=lam, nthreads=threads)
        t_acc += time.perf_counter() - t0
        pi, lam = out["pi"], out["lam"]
        solves += Bs
        steps += 1
    return {"value": solves / t_acc, "unit": "solves/s", "cores": threads, "kind": "port",
            "sample": f"{Bs} instances x {steps} RTI steps of the same workload ({t_acc:.1f} s of CPU work), "
                      f"oracle/libihm2_oracle.so with OpenMP over instances; own CPU restatement, not acados"}, check


# --------------------------------------------------------------------------------------------------------------- sub-records of the default line
CONFIG2_VARIANTS = (("fdyn6", "fdyn6", dict(track_rows="soft")),
                    ("fdyn6u", "fdyn6u", dict(terminal_bounds="stage", track_rows="soft", recover=True)),
                    ("fdyn6u_irk", "fdyn6u", dict(terminal_bounds="stage", track_rows="soft", recover=True, integrator="IRK")))
_KEEP = ("solves_per_s", "ms_per_step", "linearize_ms", "qp_ms", "ok_fraction", "status", "qp_iter_mean", "terminal_bounds", "recover", "steps", "B")


def configs2_record(B=8192, steps=20, warmup=5, device=0):
    """BASELINE.json configs[2] (batch 8192 dynamic bicycle, soft nonlinear track rows): the model as written (fdyn6, quirk Q3), the named
    deviation fdyn6u, and fdyn6u on the reference's live integrator -- solves/s, success fraction, linearise / QP ms per step."""
    var = {}
    for name, model, kw in CONFIG2_VARIANTS:
        r = rti_throughput(model=model, B=B, steps=steps, warmup=warmup, device=device, **kw)
        var[name] = {k: r[k] for k in _KEEP}
    return var


def live_options_record(B=1024, steps=20, warmup=5, persistent=True, device=0):
    """configs[1]'s batch with the solver options the reference runs live (python/main.py:227-238): SQP x 2, MERIT_BACKTRACKING, IRK GL4 x 1."""
    r = rti_throughput(model="fkin6", B=B, steps=steps, warmup=warmup, sqp="MERIT_BACKTRACKING", integrator="IRK", persistent=persistent, device=device)
    return {"control_steps_per_s": r["solves_per_s"], "ms_per_step": r["ms_per_step"], "success_fraction": r["ok_fraction"], "status_counts": r["status"],
            "qp_iter_mean_per_step": r["qp_iter_mean"], "sqp_iter_mean": r.get("sqp_iter_mean"), "alpha_mean": r.get("alpha_mean"),
            "alpha_lt1": r.get("alpha_lt1"), "steps": steps, "B": B, "persistent": persistent,
            "options": "SQP max_iter 2, MERIT_BACKTRACKING, IRK (4 Gauss-Legendre stages, 1 step), RK4 x 25 plant; status 2 = ACADOS_MAXITER, accepted (python/main.py:326)"}


# --------------------------------------------------------------------------------------------------------------- entry
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 500 for config 1, 20 for configs 2 and 3, 200 for config 4)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 50 for config 1, 5 otherwise)")
    ap.add_argument("--batch", type=int, default=None,
                    help="instances per GPU (defaults: config 1: 1024, config 2: 8192, config 3: 65536/8 = 8192, config 4: 4096/8 = 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs2 / live_options sub-records of the default line")
    ap.add_argument("--per-step-launches", action="store_true",
                    help="one launch per phase and step (ihm2mpc_step) instead of the persistent per-instance loop (ihm2mpc_run_steps)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--lib", default=None, help="path of a diagnostic build of libihm2mpc.so (tools/build_variant.sh); default: the product library")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[.]: 1 = batch 1024 kinematic (the metric's configuration); 2 = batch 8192 dynamic bicycle with "
                         "soft nonlinear track rows, as written (fdyn6) and with un-crossed slip angles (fdyn6u); 3 = 65536 dynamic bicycles over "
                         "all data/ tracks, sharded; 4 = MiL closed loop, 4096 cars x 200 steps, sharded")
    ap.add_argument("--model", default="fdyn6u", choices=["fdyn6", "fdyn6u"], help="config 3: the dynamic model (fdyn6 = as written, quirk Q3)")
    ap.add_argument("--plant-integrator", default="ERK", choices=["ERK", "IRK"],
                    help="config 4: the plant steps -- ERK = RK4 x 100 per control period, IRK = the reference's own configuration (python/main.py:395-400: "
                         "IRK, GAUSS_RADAU_IIA, 4 stages, 100 steps)")
    ap.add_argument("--live-options", action="store_true",
                    help="configs[1]'s workload with the solver options the reference runs live (python/main.py:227-238: SQP x 2 iterations, "
                         "MERIT_BACKTRACKING, IRK with 4 Gauss-Legendre stages and 1 step) instead of the metric's SQP_RTI + RK4 x 25; one GPU")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) on a multi-GPU node; gloo only to rehearse the N>1 path on one GPU")
    ap.add_argument("--device", type=int, default=None, help="HIP device ordinal of every rank (default: LOCAL_RANK); only to rehearse N > 1 on one GPU")
    ap.add_argument("--carrier", default="auto", choices=["auto", "rccl", "torch"],
                    help="who carries the barrier / max-reduce / final gather of N > 1 ranks: rccl = the C ABI's own RCCL communicator "
                         "(ihm2mpc_comm_*, no PyTorch; the 128-byte RCCL id travels over a TCP socket on MASTER_ADDR), torch = torch.distributed "
                         "with --dist-backend; auto = rccl on GPUs (backend nccl), torch for the gloo rehearsal")
    ap.add_argument("--native-rccl", action="store_true", help="the same as --carrier rccl (kept for earlier command lines)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="shard ONE global batch of this many instances over the ranks (contiguous block split, ihm2_amd/dist.py::shard_bounds: "
                         "strong scaling) instead of --batch instances per rank")
    return ap.parse_args(argv)


def claim_stdout():
    """stdout carries the JSON line and nothing else.  Libraries write to file descriptor 1 behind Python's back (RCCL's version banner on
    ncclCommInitRank, gloo's connection notes -- seen under `python -m torch.distributed.run` with two ranks): descriptor 1 is pointed at
    stderr for them, and ``sys.stdout`` keeps the original descriptor for the line."""
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(line_fd, "w")


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and (args.config == 2 or args.live_options):
        # one-GPU workloads: N processes would each run the whole of it on the same device and print N lines
        raise SystemExit("--config 2 and --live-options are one-GPU measurements: run them with --gpus 1 (configs 1, 3 and 4 shard over --gpus N)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: bench.py starts one process per GPU itself -- BEFORE anything here touches the GPU -- and rank 0 prints the line.
        # Every rank opens device LOCAL_RANK: on a node with fewer GPUs the rank fails in ihm2mpc_create and so does the run (never a silent
        # 1-GPU result); --device D puts all ranks on one device to rehearse the path.
        from ihm2_amd.dist import spawn_ranks

        raise SystemExit(spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    claim_stdout()             # after the branch above: the ranks bench.py starts inherit descriptor 1 as it was and claim it themselves
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.lib:
        from ihm2_amd import _lib

        _lib.LIB_PATH = os.path.abspath(args.lib)
    if args.config == 2:
        return main_config2(args)
    if args.live_options:
        return main_live_options(args)
    from ihm2_amd.dist import RankContext

    ranks = RankContext(carrier="rccl" if args.native_rccl else args.carrier, backend=args.dist_backend, device=args.device)
    try:
        if args.config == 3:
            main_config3(args, ranks)
        elif args.config == 4:
            main_config4(args, ranks)
        else:
            main_config1(args, ranks)
    except BaseException:
        ranks.abort()          # no barrier on the way out of a failure: the peers are in another collective, or in none (the launcher stops them)
        raise
    ranks.close()


def main_config1(args, ranks):
    from ihm2_amd.solver import BatchedOcpSolver

    steps = 500 if args.steps is None else args.steps
    warmup = 50 if args.warmup is None else args.warmup
    rank, world, device = ranks.rank, ranks.world, ranks.device
    B = 1024 if args.batch is None else args.batch
    lo = 0
    total = args.global_batch if args.global_batch else B * world
    if args.global_batch:
        lo, hi = ranks.bounds(args.global_batch)
        B = hi - lo
    ocp, track = build_problem(B)
    solver = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref, device=device)
    if args.global_batch:      # one global Monte-Carlo batch, this rank's block of it
        x0 = np.ascontiguousarray(sample_x0(track, args.global_batch, seed=20240607)[lo:lo + B])
    else:
        x0 = sample_x0(track, B, seed=20240607 + rank)
    solver.set_x0(x0)
    ranks.attach(solver, total)
    solver.init_guess()
    solver.set_lap_wrap(True)        # any number of steps: cars that complete a lap are moved back by one lap length

    SYNC_EVERY = 8
    u0_ring = solver.alloc_pinned((SYNC_EVERY, B, 2))      # pinned host memory: u0 of every step is read back in stream order

    def step(i):
        solver.step(S_TARGET, model=0, M_sim=M_SUB)        # plant + shift/ramp + one RTI iteration (ihm2mpc_step)
        solver.get_u0_async(u0_ring[i % SYNC_EVERY])       # device -> host without waiting: the next step is enqueued at once

    persistent = not args.per_step_launches
    t_lin = t_qp = t_loop = 0.0
    n_timed = 0
    if persistent:
        # ONE launch runs the K steps: every instance does plant -> shift/ramp -> linearise -> QP step after step on its own
        # wavefront (no instance waits for the slowest QP of the batch); u0 and the QP iteration count of every step come back
        # to pinned host memory inside the timed region
        n_hist = max(steps, warmup, 1)
        u0_hist = solver.alloc_pinned((n_hist, B, 2))
        it_hist = np.zeros((n_hist, B), dtype=np.int32)
        solver.reserve_history(n_hist)
        solver.step(S_TARGET, model=0, M_sim=M_SUB)        # a first solve (untimed, launched per phase): the loop starts from a control
        if warmup:
            solver.run_steps(S_TARGET, warmup, model=0, M_sim=M_SUB, u0_hist=u0_hist[:warmup])
        ranks.barrier(solver)
        t0 = time.perf_counter()
        solver.run_steps(S_TARGET, steps, model=0, M_sim=M_SUB, u0_hist=u0_hist[:steps], qp_iter_hist=it_hist[:steps])
        ranks.barrier(solver)
        elapsed = time.perf_counter() - t0
        t_loop = solver.get_timings()["total_ms"]          # HIP events around the launch, on the solver's stream
        qp_iters_timed = it_hist[:steps]
    else:
        for i in range(warmup):
            step(i)
        ranks.barrier(solver)
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
            if i % SYNC_EVERY == SYNC_EVERY - 1 or i == steps - 1:
                solver.synchronize()           # the host runs at most SYNC_EVERY steps ahead of the device
                tm = solver.get_timings()      # HIP events recorded on the solver's stream around the kernels of this step
                t_lin += tm["linearize_ms"]; t_qp += tm["qp_ms"]; n_timed += 1
        ranks.barrier(solver)
        elapsed = time.perf_counter() - t0
        qp_iters_timed = None
    # per-step latency of a batch (step + read-back + wait), outside the timed region: 24 synchronous steps
    step_ms = []
    t_lin_s = t_qp_s = 0.0
    for i in range(24):
        ts = time.perf_counter()
        step(i); solver.synchronize()
        step_ms.append((time.perf_counter() - ts) * 1e3)
        tm = solver.get_timings(); t_lin_s += tm["linearize_ms"] / 24; t_qp_s += tm["qp_ms"] / 24
    qp_iters = solver.get_qp_iter() if qp_iters_timed is None else qp_iters_timed
    # per-solve latency of ONE instance (B = 1: the real-time controller of mpc_control_node.cpp), rank 0 only: x0 in from the
    # host, one control step, u0 back on the host, 60 synchronous steps after 10 of warm-up
    b1_ms = []
    if rank == 0:
        one = BatchedOcpSolver(ocp, 1, track.s_ref, track.kappa_ref, device=device)
        one.set_x0(x0[:1]); one.init_guess()
        for i in range(70):
            ts = time.perf_counter()
            one.sim_advance(model=0, M_sim=M_SUB)
            # the state crosses the boundary as a host buffer, as in the ROS node: x0 in, u0 out, one call (ihm2mpc_compute_control)
            one.compute_control(one.get_x0(), S_TARGET)
            if i >= 10:
                b1_ms.append((time.perf_counter() - ts) * 1e3)
        one.free()

    elapsed = ranks.max(elapsed)
    tg0 = time.perf_counter()
    u0_all, status_all = ranks.gather_results(solver)          # the final gather of (u0, status) in global order
    gather_ms = (time.perf_counter() - tg0) * 1e3 if world > 1 or ranks.carrier != "none" else 0.0
    assert u0_all.shape == (total, 2) and status_all.shape == (total,)
    collective = ranks.verify(solver)          # rank count of the communicator + every rank's device: n_gpus = N only if N distinct devices answered
    ranks.detach()                             # the last exchange is done: the handle's communicator goes before any solver is freed

    if rank == 0:
        total_solves = total * steps
        value = total_solves / elapsed
        n_ipm = float(np.mean(qp_iters))
        f_lin, f_qp = flops_per_solve(n_ipm)
        if persistent:      # the dominant (only) kernel of the timed region: one launch = steps x batch solves
            ms_lin, ms_qp = t_lin_s, t_qp_s      # per-phase split: from the per-step launches of the latency section below
            kname, kms, kflops = "k_steps", t_loop, (f_lin + f_qp) * B * steps
        else:
            ms_lin, ms_qp = t_lin / max(n_timed, 1), t_qp / max(n_timed, 1)
            if ms_lin >= ms_qp:
                kname, kms, kflops = "k_linearize", ms_lin, f_lin * B
            else:
                kname, kms, kflops = "k_qp_wave", ms_qp, f_qp * B
        achieved = kflops / (kms * 1e-3) / 1e12
        alg_bytes = 8 * (2 * (N_H + 1) * 8 + 2 * N_H * 2 + 8 + 5)          # 6632 B per solve
        # HBM traffic per launch of the dominant kernel: PMC counters (FETCH_SIZE, WRITE_SIZE) cannot be read from inside this
        # process; the per-solve figure comes from the committed rocprofv3 --pmc passes of `bench.py --steps 20 --warmup 5`
        # (profiles/r*/bench_traffic.json, corrected as MI355X_MICROARCH.md prescribes) and is scaled by the solves of this launch
        traffic, traffic_src = None, None
        for cand in (TRAFFIC_JSON, os.path.join("profiles", "r3", "bench_traffic.json")):
            try:
                tj = json.load(open(os.path.join(ROOT, cand)))
                key = "k_steps" if persistent else kname
                if B == tj["batch"] and key in tj["bytes_per_solve"]:
                    traffic = tj["bytes_per_solve"][key] * B * (steps if persistent else 1)
                    traffic_src = f"{cand} ({tj['command']}): {tj['bytes_per_solve'][key]:.0f} B per solve x solves of this launch"
                    break
            except Exception:
                pass
        # KKT residuals of the QPs of the last step at the points the solver returned (inf-norms, relative to the scales the
        # interior-point tolerance 1e-6 is taken against; max over the instances of this rank) ...
        qres = solver.get_qp_residuals()
        okq = solver.get_status() == 0              # the residuals of a failed QP are what it stopped at, not a solution's
        qr = qres[okq] if okq.any() else np.full((1, 4), np.nan)
        kkt = {"stat": float(qr[:, 0].max()), "eq": float(qr[:, 1].max()), "ineq": float(qr[:, 2].max()), "comp": float(qr[:, 3].max()),
               "tolerance": float(ocp.solver_options.qp_tol), "relative": True, "over_status_0_fraction": float(okq.mean())}
        # ... and the NLP residuals at the iterates those QPs were built at (what acados' get_stats("residuals") reports in RTI mode:
        # one SQP iteration per control step does not converge the NLP, the reference ramp moves it every step)
        res_all = solver.get_residuals()
        nlp_res = {"stat": float(res_all[:, 0].max()), "eq": float(res_all[:, 1].max()), "ineq": float(res_all[:, 2].max()), "comp": float(res_all[:, 3].max())}
        tot20 = None
        if persistent and steps >= 2:        # how far the launch's slowest instance lies above the mean (the launch ends with it)
            tot = qp_iters.sum(axis=0)
            tot20 = {"mean": float(tot.mean()), "p99": float(np.percentile(tot, 99)), "max": int(tot.max())}
        out = {
            "metric": "NMPC RTI solves/s (batch), N=40, nx=8 (6-DOF bicycle), fkin6",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: batch={B}/GPU kinematic bicycle fkin6, N=40, dt=0.05, RK4 x M={M_SUB}, "
                                   f"track {TRACK}, closed-loop step = plant + shift/ramp + 1 SQP-RTI iteration + u0 read-back to pinned host memory; "
                                   + ("all steps in one launch, every instance on its own wavefront (ihm2mpc_run_steps)" if persistent else "one launch per phase and step (ihm2mpc_step)"),
                       "batch_per_gpu": B, "global_batch": total, "N": N_H, "M": M_SUB, "parallelism": f"{world} x independent shards",
                       "collective": collective},
            "roofline": {"bound": "fp64_valu", "kernel": kname, "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes * B * (steps if persistent else 1),
                         "launch": f"{steps} control steps x {B} instances in one launch (persistent per-instance loop)" if persistent else "one RTI iteration of the batch",
                         "note": "fp64 path bound by instruction issue / latency of one wave per SIMD; linearisation on the fp64 vector "
                                 "pipe, the QP's Riccati factor on v_mfma_f64_16x16x4_f64; peak = MI355X fp64 vector = fp64 matrix peak; "
                                 "achieved = algorithmic flops (SURVEY.md 8d model) / HIP-event kernel time",
                         "kernel_ms": kms, "linearize_ms": ms_lin, "qp_ms": ms_qp, "n_ipm_mean": n_ipm, "M": M_SUB,
                         "alg_flops_per_solve": f_lin + f_qp,
                         "hbm_fraction_algorithmic": alg_bytes * value / world / 1e9 / HBM_PEAK_GBS,
                         "hbm_fraction_executed": (traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         # where the gap is: the two phases of a step as their own kernels (the per-step launches of the latency section below;
                         # algorithmic flops of the phase / its kernel time / peak), and the executed traffic against the algorithmic bytes
                         "phases": {"linearize": {"ms": ms_lin, "tflops": f_lin * B / (ms_lin * 1e-3) / 1e12 if ms_lin else None,
                                                  "frac": f_lin * B / (ms_lin * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if ms_lin else None},
                                    "qp": {"ms": ms_qp, "tflops": f_qp * B / (ms_qp * 1e-3) / 1e12 if ms_qp else None,
                                           "frac": f_qp * B / (ms_qp * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if ms_qp else None}},
                         "wasted_traffic_ratio": (traffic / (alg_bytes * B * (steps if persistent else 1))) if traffic else None},
            "latency_ms_p50_batch": float(np.percentile(step_ms, 50)), "latency_ms_p99_batch": float(np.percentile(step_ms, 99)),
            "latency_ms_p50_single": float(np.percentile(b1_ms, 50)), "latency_ms_p99_single": float(np.percentile(b1_ms, 99)),
            "status_counts": {str(k): int(v) for k, v in enumerate(np.bincount(status_all, minlength=5)) if v},
            "gather_ms": gather_ms,
            "max_kkt_residual": kkt, "nlp_residuals_at_linearisation_point": nlp_res,
            "qp_iter_percentiles": {"p50": float(np.percentile(qp_iters, 50)), "p90": float(np.percentile(qp_iters, 90)),
                                    "p99": float(np.percentile(qp_iters, 99)), "max": int(np.max(qp_iters))},
            "qp_iter_per_instance_over_launch": tot20,
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is timed at N = 1 only
            ns = min(64, B)
            pi_g, lam_g = solver.get_multipliers()
            state = (solver.get_x()[:ns], solver.get_u()[:ns], solver.get_x0()[:ns], pi_g[:ns], lam_g[:ns])
            out["cpu_baseline"], chk = cpu_baseline(ocp, track, x0, args.cpu_budget, gpu_state=state, device=device)
            out["max_rel_dev_vs_oracle"] = max(chk["max_rel_dev_x"], chk["max_rel_dev_u"])
            out["oracle_check"] = chk
        if not args.no_extras and world == 1 and not args.global_batch:
            # short runs of BASELINE.json configs[2] and of the reference's live options, after (outside) the headline's timed region
            solver.free(); solver = None
            out["configs2"] = configs2_record(device=device)
            out["live_options"] = live_options_record(device=device)
        print(json.dumps(out), flush=True)
    if solver is not None:
        ranks.barrier(solver)
        solver.free()


def main_config2(args):
    """BASELINE.json configs[2]: batch 8192, dynamic bicycle (Pacejka tyres), N = 40, soft nonlinear track rows, one GPU.
    Launches per phase and step: at this batch they beat the one-launch loop of the dynamic models (k_steps<..,DYN=1>, eight rounds of whole
    histories; measured round 4 with failed instances re-initialised between steps only on this path: 323 k against 292 k solves/s with RK4 x 25,
    422 k against 368 k with the collocation integrator, profiles/r4/dyn_persistent_vs_perstep.jsonl).  Both the model as written (quirk Q3:
    crossed slip angles, open-loop unstable -- most QPs are infeasible) and the named deviation fdyn6u are run and reported."""
    B = 8192 if args.batch is None else args.batch
    steps = 20 if args.steps is None else min(args.steps, 50)
    warmup = 5 if args.warmup is None else min(args.warmup, 10)
    var = configs2_record(B=B, steps=steps, warmup=warmup, device=args.device or 0)
    r = var["fdyn6"]
    print(json.dumps({
        "metric": "NMPC RTI solves/s (batch), N=40, nx=8 (6-DOF dynamic bicycle, Pacejka), fdyn6 + soft track rows",
        "value": r["solves_per_s"], "unit": "solves/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"configs[2]: batch={B} dynamic bicycle fdyn6 as written (python/models.py:455-606), N=40, dt=0.05, RK4 x M=25, soft nonlinear "
                               "track rows 100/100 (old/generate_acaods_interface.py:191-212,380-449), track fsds_competition_1; launches per phase and step",
                   "batch_per_gpu": B, "N": N_H, "M": M_SUB},
        "success_fraction": r["ok_fraction"],
        "variants": var,
        "note": "value = the model as written; its QPs are mostly infeasible (open-loop unstable, DESIGN.md section 2). variants.fdyn6u = "
                "un-crossed slip angles (named deviation), stage terminal box, re-initialisation of failed instances; variants.fdyn6u_irk = the same "
                "with the reference's live integrator (python/main.py:234-236: IRK, 4 Gauss-Legendre stages, 1 step) instead of RK4 x 25"}))


def main_config3(args, ranks):
    """BASELINE.json configs[3]: ONE Monte-Carlo batch of dynamic bicycles over all seven data/ tracks (instance g on track g mod 7), soft
    nonlinear track rows, sharded over the ranks by contiguous blocks; the only exchange is the final gather of (u0, status).
    Default size: 8192 instances per GPU (N = 8 gives the 65 536 of the configuration); --global-batch fixes the total instead."""
    from ihm2_amd.track import track_table

    steps = 20 if args.steps is None else args.steps
    warmup = 5 if args.warmup is None else args.warmup
    per_gpu = CONFIG3_TOTAL // NODE_GPUS if args.batch is None else args.batch
    total = args.global_batch if args.global_batch else per_gpu * ranks.world
    lo, hi = ranks.bounds(total)
    plans = [track_table(t) for t in ALL_TRACKS]
    x0_all, tid_all = monte_carlo_batch(plans, total)
    ranks.total = total
    kw = dict(terminal_bounds="stage", track_rows="soft", recover=True) if args.model == "fdyn6u" else dict(track_rows="soft")
    r = rti_throughput(model=args.model, B=hi - lo, steps=steps, warmup=warmup, tracks=ALL_TRACKS, device=ranks.device,
                       x0=np.ascontiguousarray(x0_all[lo:hi]), track_id=tid_all[lo:hi], ranks=ranks, **kw)
    elapsed, n_ok = r["elapsed_max_s"], r["n_ok_sum"]
    u0_all, st_all = r["gathered"]
    assert u0_all.shape == (total, 2) and st_all.shape == (total,)
    if ranks.rank == 0:
        print(json.dumps({
            "metric": f"NMPC RTI solves/s (batch), N=40, nx=8 (6-DOF dynamic bicycle, Pacejka), {args.model} + soft track rows, all data/ tracks",
            "value": total * steps / elapsed, "unit": "solves/s", "n_gpus": ranks.world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[3]: global batch {total} dynamic bicycles ({args.model}), instance g on track g mod 7 over {', '.join(ALL_TRACKS)}, N=40, dt=0.05, "
                                   "RK4 x M=25, soft nonlinear track rows 100/100, contiguous block split over the ranks (shard_bounds); launches per phase and step",
                       "batch_per_gpu": hi - lo, "global_batch": total, "N": N_H, "M": M_SUB, "parallelism": f"{ranks.world} x independent shards",
                       "collective": r["collective"]},
            "gather_ms": r["gather_ms"],
            "success_fraction": n_ok / (total * steps),
            "status_counts_last_step": {str(k): int(v) for k, v in enumerate(np.bincount(st_all, minlength=5)) if v},
            "rank0": {k: r[k] for k in _KEEP},
            "note": "fdyn6u = the dynamic model with un-crossed slip angles (named deviation from quirk Q3, DESIGN.md section 2), stage terminal box, failed "
                    "instances re-initialised; --model fdyn6 runs the model as written"}), flush=True)


def main_config4(args, ranks):
    """BASELINE.json configs[4]: the MiL loop (new_python/closed_loop_sim.py:34-42; python/main.py:448-517) for a batch of cars -- controller and
    plant of a car on the same GPU, cars sharded over the ranks by contiguous blocks.  Default: 512 cars per GPU (N = 8 gives the 4096 of the
    configuration); --global-batch fixes the total instead.  Up to 1024 cars per GPU every car runs its 200 control periods back to back on its
    own wavefront in one launch (ihm2mpc_run_steps with the freezing rules on the device); larger shares take one ihm2mpc_step per period."""
    from ihm2_amd.track import track_table

    steps = 200 if args.steps is None else args.steps
    per_gpu = CONFIG4_TOTAL // NODE_GPUS if args.batch is None else args.batch
    total = args.global_batch if args.global_batch else per_gpu * ranks.world
    lo, hi = ranks.bounds(total)
    plan = track_table(TRACK)
    x0_all = sample_x0(plan, total, seed=5)
    ranks.total = total
    B = hi - lo
    persistent = B <= 1024 and not args.per_step_launches
    r = closed_loop_config5(B=B, steps=steps, terminal_bounds="stage", plant="KIN6_DYN6U", soft_state_bounds=(1000.0, 1000.0), device_loop=not persistent,
                            persistent=persistent, device=ranks.device, x0=np.ascontiguousarray(x0_all[lo:hi]), ranks=ranks, plant_integrator=args.plant_integrator)
    elapsed = r["elapsed_max_s"]
    alive, finished, failed = (r[k + "_sum"] for k in ("alive", "finished", "failed"))
    u0_all, st_all = r["gathered"]
    assert u0_all.shape == (total, 2)
    as_written = None
    if ranks.world == 1 and not args.no_extras:
        # the configuration as the reference writes it -- plant KIN6_DYN6 (crossed slip angles, quirk Q3), reference terminal box, hard boxes --
        # on a sample of this rank's cars, outside the timed region: what becomes of the cars is part of the line
        ns = min(B, 256)
        w = closed_loop_config5(B=ns, steps=steps, terminal_bounds="reference", plant="KIN6_DYN6", soft_state_bounds=None, device_loop=True, device=ranks.device,
                                x0=np.ascontiguousarray(x0_all[lo:lo + ns]), plant_integrator=args.plant_integrator)
        as_written = {"cars": ns, "alive": w["alive"], "finished": w["finished"], "failed_or_stopped": ns - w["alive"] - w["finished"], "steps_run": w["steps"],
                      "control_steps_per_s": w["control_steps_per_s"],
                      "note": "plant KIN6_DYN6 as written (python/models.py:543-546 feeds every wheel the slip angle of the opposite one: open-loop unstable), "
                              "reference terminal box, hard state boxes: the reference's stop rule (python/main.py:326-328) ends nearly every run in the first corner"}
    plant_txt = "RK4 x 100" if args.plant_integrator == "ERK" else "IRK Radau IIA, 4 stages x 100 steps (python/main.py:395-400)"
    if ranks.rank == 0:
        print(json.dumps({
            "metric": "MiL control steps/s (cars x control periods / wall time), NMPC fkin6 N=40 + plant KIN6_DYN6U = NAMED DEVIATION from the reference's "
                      f"KIN6_DYN6 (un-crossed slip angles; stage terminal box; soft plant-state bounds), {plant_txt}",
            "value": total * steps / elapsed, "unit": "control steps/s", "n_gpus": ranks.world, "steps": steps, "warmup": 0, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[4]: {total} cars x {steps} control periods, controller = SQP-RTI NMPC (fkin6, N=40, stage terminal box, soft plant-state "
                                   f"bounds), plant = kinematic / un-crossed dynamic switch of python/main.py:482-489 ({plant_txt}), freezing rules of python/main.py:326-328,"
                                   "503-504,514-517 on the device; " + ("one launch per car history (ihm2mpc_run_steps)" if persistent else "one ihm2mpc_step per period"),
                       "batch_per_gpu": B, "global_batch": total, "N": N_H, "parallelism": f"{ranks.world} x independent shards", "collective": r["collective"]},
            "gather_ms": r["gather_ms"],
            "cars_alive": alive, "cars_finished": finished, "cars_failed": failed, "plant_integrator": args.plant_integrator,
            "reference_configuration_KIN6_DYN6": as_written,
            "rank0": {k: r[k] for k in ("control_steps_per_s", "alive", "finished", "failed", "mean_speed", "progress_m_median", "persistent", "B")}}), flush=True)


def main_live_options(args):
    """configs[1]'s batch with the reference's live solver options (python/main.py:227-238) instead of the metric's SQP_RTI + RK4: SQP with
    two iterations, merit backtracking and the 4-stage Gauss-Legendre collocation step -- all control steps in one launch (IRK and the line
    search inside the persistent loop) unless --per-step-launches."""
    B = 1024 if args.batch is None else args.batch
    steps = 500 if args.steps is None else args.steps
    warmup = 50 if args.warmup is None else args.warmup
    r = live_options_record(B=B, steps=steps, warmup=warmup, persistent=not args.per_step_launches, device=args.device or 0)
    print(json.dumps({
        "metric": "NMPC control steps/s (batch), live solver options: SQP x 2, MERIT_BACKTRACKING, IRK GL4 x 1, N=40, nx=8, fkin6",
        "value": r["control_steps_per_s"], "unit": "control steps/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"configs[1] with the live options of python/main.py:227-238: batch={B} kinematic bicycle fkin6, N=40, dt=0.05, SQP max_iter 2, "
                               "MERIT_BACKTRACKING, IRK (4 Gauss-Legendre stages, 1 step per interval), RK4 x 25 plant, track fsds_competition_1; "
                               + ("launches per phase and step" if args.per_step_launches else "all steps in one launch (ihm2mpc_run_steps)"),
                   "batch_per_gpu": B, "N": N_H},
        "success_fraction": r["success_fraction"], "status_counts": r["status_counts"], "qp_iter_mean_per_step": r["qp_iter_mean_per_step"],
        "sqp_iter_mean": r.get("sqp_iter_mean"), "alpha_mean": r.get("alpha_mean"), "alpha_lt1": r.get("alpha_lt1"),
        "note": "status 2 = ACADOS_MAXITER after the two iterations, which the reference accepts (python/main.py:326); not the headline metric "
                "(that is the default run: SQP_RTI, RK4 x 25)"}))


if __name__ == "__main__":
    main()
