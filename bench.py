#!/usr/bin/env python3
"""bench.py -- NMPC RTI solves/s of the batched bicycle NMPC on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], per GPU): batch = 1024 kinematic bicycle (fkin6) instances, N = 40,
dt = 0.05, RK4 x M = 25, track fsds_competition_1, synthetic Monte-Carlo initial states of SURVEY.md 8d.
One "step" = one closed-loop control step of the whole batch, everything resident on the device:
plant advance (kinematic model, RK4 x 25) -> reference ramp + warm-start shift (python/main.py:303-322)
-> one SQP-RTI iteration (linearise, QP, full step) -> read back u0 (B x 2) to the host.
Weak scaling: every rank owns its own 1024 instances on its own GPU; no data-path collective; RCCL is
used for the barrier / max-reduce of the timing and for one final all_gather of the results.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (live HIP-event time on the
solver's stream); `cpu_baseline` is the CPU oracle (own restatement, NOT acados) on the host cores.
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
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector peak = fp64 matrix peak (guide: MI355X_MICROARCH.md / SURVEY.md 8d)
HBM_PEAK_GBS = 8000.0


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


def flops_per_solve(n_ipm):
    """Algorithmic flop model of SURVEY.md 8d (fkin6): linearisation + structure-exploiting QP."""
    nx, nu = 8, 2
    lin = N_H * M_SUB * 4 * (2 * nx * nx * (nx + nu) + 400)
    qp = n_ipm * N_H * 1800.0
    return lin, qp


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU (BASELINE configs[1]: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-step-launches", action="store_true",
                    help="one launch per phase and step (ihm2mpc_step) instead of the persistent per-instance loop (ihm2mpc_run_steps)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--lib", default=None, help="path of a diagnostic build of libihm2mpc.so (tools/build_variant.sh); default: the product library")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2],
                    help="BASELINE.json configs[.]: 1 = batch 1024 kinematic (the metric's configuration); 2 = batch 8192 dynamic bicycle with "
                         "soft nonlinear track rows, reported as written (fdyn6) and with un-crossed slip angles (fdyn6u)")
    ap.add_argument("--live-options", action="store_true",
                    help="configs[1]'s workload with the solver options the reference runs live (python/main.py:227-238: SQP x 2 iterations, "
                         "MERIT_BACKTRACKING, IRK with 4 Gauss-Legendre stages and 1 step) instead of the metric's SQP_RTI + RK4 x 25; one GPU")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) on a multi-GPU node; gloo only to rehearse the N>1 path on one GPU")
    ap.add_argument("--device", type=int, default=None, help="HIP device ordinal (default: LOCAL_RANK)")
    ap.add_argument("--native-rccl", action="store_true",
                    help="barrier / max-reduce / final gather over the C ABI's own RCCL communicator (ihm2mpc_comm_*, no PyTorch) instead of "
                         "torch.distributed; the RCCL id travels over a TCP socket on MASTER_ADDR")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="shard ONE global batch of this many instances over the ranks (contiguous block split, ihm2_amd/dist.py::shard_bounds: "
                         "strong scaling, BASELINE.json configs[3] style) instead of --batch instances per rank")
    args = ap.parse_args()
    if args.lib:
        from ihm2_amd import _lib

        _lib.LIB_PATH = os.path.abspath(args.lib)
    if args.config == 2:
        return main_config2(args)
    if args.live_options:
        return main_live_options(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    use_dist = world > 1 or os.environ.get("IHM2_FORCE_DIST") == "1"      # the latter: rehearse the RCCL path on one GPU
    native = None
    if use_dist and not args.native_rccl:
        import torch
        import torch.distributed as dist

        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    from ihm2_amd.solver import BatchedOcpSolver

    B = args.batch
    lo = 0
    if args.global_batch:
        from ihm2_amd.dist import shard_bounds

        lo, hi = shard_bounds(args.global_batch, world, rank)
        B = hi - lo
    ocp, track = build_problem(B)
    device = local_rank if args.device is None else args.device
    on_gpu = args.dist_backend == "nccl"
    solver = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref, device=device)
    if args.global_batch:      # one global Monte-Carlo batch, this rank's block of it
        x0 = np.ascontiguousarray(sample_x0(track, args.global_batch, seed=20240607)[lo:lo + B])
    else:
        x0 = sample_x0(track, B, seed=20240607 + rank)
    solver.set_x0(x0)
    if use_dist and args.native_rccl:
        from ihm2_amd.dist import NativeComm

        total = args.global_batch if args.global_batch else B * world
        native = NativeComm(solver, total, world, rank, os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + 23)
    solver.init_guess()
    solver.set_lap_wrap(True)        # any number of steps: cars that complete a lap are moved back by one lap length

    SYNC_EVERY = 8
    u0_ring = solver.alloc_pinned((SYNC_EVERY, B, 2))      # pinned host memory: u0 of every step is read back in stream order

    def step(i):
        solver.step(S_TARGET, model=0, M_sim=M_SUB)        # plant + shift/ramp + one RTI iteration (ihm2mpc_step)
        solver.get_u0_async(u0_ring[i % SYNC_EVERY])       # device -> host without waiting: the next step is enqueued at once

    def barrier():
        solver.synchronize()
        if native is not None:
            native.allreduce_max(0.0)
        if dist is not None:
            import torch

            if on_gpu:
                torch.cuda.synchronize()
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()

    persistent = not args.per_step_launches
    t_lin = t_qp = t_loop = 0.0
    n_timed = 0
    if persistent:
        # ONE launch runs the K steps: every instance does plant -> shift/ramp -> linearise -> QP step after step on its own
        # wavefront (no instance waits for the slowest QP of the batch); u0 and the QP iteration count of every step come back
        # to pinned host memory inside the timed region
        n_hist = max(args.steps, args.warmup, 1)
        u0_hist = solver.alloc_pinned((n_hist, B, 2))
        it_hist = np.zeros((n_hist, B), dtype=np.int32)
        solver.reserve_history(n_hist)
        solver.step(S_TARGET, model=0, M_sim=M_SUB)        # a first solve (untimed, launched per phase): the loop starts from a control
        if args.warmup:
            solver.run_steps(S_TARGET, args.warmup, model=0, M_sim=M_SUB, u0_hist=u0_hist[:args.warmup])
        barrier()
        t0 = time.perf_counter()
        solver.run_steps(S_TARGET, args.steps, model=0, M_sim=M_SUB, u0_hist=u0_hist[:args.steps], qp_iter_hist=it_hist[:args.steps])
        barrier()
        elapsed = time.perf_counter() - t0
        t_loop = solver.get_timings()["total_ms"]          # HIP events around the launch, on the solver's stream
        qp_iters_timed = it_hist[:args.steps]
    else:
        for i in range(args.warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
            if i % SYNC_EVERY == SYNC_EVERY - 1 or i == args.steps - 1:
                solver.synchronize()           # the host runs at most SYNC_EVERY steps ahead of the device
                tm = solver.get_timings()      # HIP events recorded on the solver's stream around the kernels of this step
                t_lin += tm["linearize_ms"]; t_qp += tm["qp_ms"]; n_timed += 1
        barrier()
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
    status = solver.get_status()
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

    if native is not None:
        elapsed = native.allreduce_max(elapsed)
        tg0 = time.perf_counter()
        u0_all, status_all = native.allgather_results()      # RCCL all-gather of (u0, status) behind the C ABI, host copies on every rank
        gather_ms = (time.perf_counter() - tg0) * 1e3
        assert u0_all.shape == (native.total, 2)
    elif dist is not None:
        import torch

        from ihm2_amd.dist import all_gather_blocks

        dev = "cuda" if on_gpu else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # final gather of the results over RCCL/xGMI: u0 (B,2) and status of every rank, device to device
        tg0 = time.perf_counter()
        if on_gpu:
            u0_dev = torch.empty((B, 2), dtype=torch.float64, device="cuda")
            st_dev = torch.empty((B,), dtype=torch.int32, device="cuda")
            solver.get_u0_device(u0_dev.data_ptr()); solver.get_status_device(st_dev.data_ptr()); solver.synchronize()
        else:
            u0_dev = torch.from_numpy(solver.get_u0()); st_dev = torch.from_numpy(status.copy())
        u0_all = all_gather_blocks(u0_dev, B * world)
        st_all = [all_gather_blocks(st_dev, B * world)]
        if on_gpu:
            torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg0) * 1e3
        assert u0_all.shape == (B * world, 2)
        status_all = torch.cat(st_all).cpu().numpy()
    else:
        gather_ms = 0.0
        status_all = status

    if rank == 0:
        total_solves = (args.global_batch if args.global_batch else B * world) * args.steps
        value = total_solves / elapsed
        n_ipm = float(np.mean(qp_iters))
        f_lin, f_qp = flops_per_solve(n_ipm)
        if persistent:      # the dominant (only) kernel of the timed region: one launch = steps x batch solves
            ms_lin, ms_qp = t_lin_s, t_qp_s      # per-phase split: from the per-step launches of the latency section below
            kname, kms, kflops = "k_steps", t_loop, (f_lin + f_qp) * B * args.steps
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
        # (profiles/r2/bench_traffic.json, corrected as MI355X_MICROARCH.md prescribes) and is scaled by the solves of this launch
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r2", "bench_traffic.json")))
            key = "k_steps" if persistent else kname
            if B == tj["batch"] and key in tj["bytes_per_solve"]:
                traffic = tj["bytes_per_solve"][key] * B * (args.steps if persistent else 1)
                traffic_src = f"profiles/r2/bench_traffic.json ({tj['command']}): {tj['bytes_per_solve'][key]:.0f} B per solve x solves of this launch"
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
        out = {
            "metric": "NMPC RTI solves/s (batch), N=40, nx=8 (6-DOF bicycle), fkin6",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: batch={B}/GPU kinematic bicycle fkin6, N=40, dt=0.05, RK4 x M={M_SUB}, "
                                   f"track {TRACK}, closed-loop step = plant + shift/ramp + 1 SQP-RTI iteration + u0 read-back to pinned host memory; "
                                   + ("all steps in one launch, every instance on its own wavefront (ihm2mpc_run_steps)" if persistent else "one launch per phase and step (ihm2mpc_step)"),
                       "batch_per_gpu": B, "N": N_H, "M": M_SUB, "parallelism": f"{world} x independent shards",
                       "collective": ("RCCL all-gather of (u0, status) behind the C ABI (ihm2mpc_comm_*)" if native is not None else
                                      "torch.distributed all_gather of (u0, status)" if dist is not None else "none (one GPU)")},
            "roofline": {"bound": "fp64_valu", "kernel": kname, "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes * B * (args.steps if persistent else 1),
                         "launch": f"{args.steps} control steps x {B} instances in one launch (persistent per-instance loop)" if persistent else "one RTI iteration of the batch",
                         "note": "fp64 path bound by instruction issue / latency of one wave per SIMD; linearisation on the fp64 vector "
                                 "pipe, the QP's Riccati factor on v_mfma_f64_16x16x4_f64; peak = MI355X fp64 vector = fp64 matrix peak; "
                                 "achieved = algorithmic flops (SURVEY.md 8d model) / HIP-event kernel time",
                         "kernel_ms": kms, "linearize_ms": ms_lin, "qp_ms": ms_qp, "n_ipm_mean": n_ipm, "M": M_SUB,
                         "alg_flops_per_solve": f_lin + f_qp,
                         "hbm_fraction_algorithmic": alg_bytes * value / world / 1e9 / HBM_PEAK_GBS,
                         "hbm_fraction_executed": (traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None},
            "latency_ms_p50_batch": float(np.percentile(step_ms, 50)), "latency_ms_p99_batch": float(np.percentile(step_ms, 99)),
            "latency_ms_p50_single": float(np.percentile(b1_ms, 50)), "latency_ms_p99_single": float(np.percentile(b1_ms, 99)),
            "status_counts": {str(k): int(v) for k, v in enumerate(np.bincount(status_all, minlength=5)) if v},
            "gather_ms": gather_ms,
            "max_kkt_residual": kkt, "nlp_residuals_at_linearisation_point": nlp_res,
            "qp_iter_percentiles": {"p50": float(np.percentile(qp_iters, 50)), "p90": float(np.percentile(qp_iters, 90)),
                                    "p99": float(np.percentile(qp_iters, 99)), "max": int(np.max(qp_iters))},
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is timed at N = 1 only
            ns = min(64, B)
            pi_g, lam_g = solver.get_multipliers()
            state = (solver.get_x()[:ns], solver.get_u()[:ns], solver.get_x0()[:ns], pi_g[:ns], lam_g[:ns])
            out["cpu_baseline"], chk = cpu_baseline(ocp, track, x0, args.cpu_budget, gpu_state=state, device=device)
            out["max_rel_dev_vs_oracle"] = max(chk["max_rel_dev_x"], chk["max_rel_dev_u"])
            out["oracle_check"] = chk
        print(json.dumps(out))
    if native is not None:
        native.allreduce_max(0.0)
        native.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    solver.free()


def main_config2(args):
    """BASELINE.json configs[2]: batch 8192, dynamic bicycle (Pacejka tyres), N = 40, soft nonlinear track rows, one GPU.
    Launches per phase and step (the persistent loop has no dynamic-model instantiation).  Both the model as written (quirk Q3:
    crossed slip angles, open-loop unstable -- most QPs are infeasible) and the named deviation fdyn6u are run and reported."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from measure_configs import rti_throughput

    B = args.batch if args.batch != 1024 else 8192
    steps, warmup = min(args.steps, 50), min(args.warmup, 10)
    var = {}
    for name, model, kw in (("fdyn6", "fdyn6", dict(track_rows="soft")),
                            ("fdyn6u", "fdyn6u", dict(terminal_bounds="stage", track_rows="soft", recover=True)),
                            ("fdyn6u_irk", "fdyn6u", dict(terminal_bounds="stage", track_rows="soft", recover=True, integrator="IRK"))):
        r = rti_throughput(model=model, B=B, steps=steps, warmup=warmup, **kw)
        var[name] = {k: r[k] for k in ("solves_per_s", "ms_per_step", "linearize_ms", "qp_ms", "ok_fraction", "status", "qp_iter_mean", "terminal_bounds", "recover")}
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


def main_live_options(args):
    """configs[1]'s batch with the reference's live solver options (python/main.py:227-238) instead of the metric's SQP_RTI + RK4: SQP with
    two iterations, merit backtracking and the 4-stage Gauss-Legendre collocation step -- all control steps in one launch (IRK and the line
    search inside the persistent loop) unless --per-step-launches."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from measure_configs import rti_throughput

    B = args.batch
    r = rti_throughput(model="fkin6", B=B, steps=args.steps, warmup=args.warmup, sqp="MERIT_BACKTRACKING", integrator="IRK", persistent=not args.per_step_launches)
    print(json.dumps({
        "metric": "NMPC control steps/s (batch), live solver options: SQP x 2, MERIT_BACKTRACKING, IRK GL4 x 1, N=40, nx=8, fkin6",
        "value": r["solves_per_s"], "unit": "control steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"configs[1] with the live options of python/main.py:227-238: batch={B} kinematic bicycle fkin6, N=40, dt=0.05, SQP max_iter 2, "
                               "MERIT_BACKTRACKING, IRK (4 Gauss-Legendre stages, 1 step per interval), RK4 x 25 plant, track fsds_competition_1; "
                               + ("launches per phase and step" if args.per_step_launches else "all steps in one launch (ihm2mpc_run_steps)"),
                   "batch_per_gpu": B, "N": N_H},
        "success_fraction": r["ok_fraction"], "status_counts": r["status"], "qp_iter_mean_per_step": r["qp_iter_mean"],
        "sqp_iter_mean": r.get("sqp_iter_mean"), "alpha_mean": r.get("alpha_mean"), "alpha_lt1": r.get("alpha_lt1"),
        "note": "status 2 = ACADOS_MAXITER after the two iterations, which the reference accepts (python/main.py:326); not the headline metric "
                "(that is the default run: SQP_RTI, RK4 x 25)"}))


if __name__ == "__main__":
    main()
