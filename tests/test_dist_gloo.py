"""N > 1 path on CPU: two gloo ranks shard a batch, each solves its own block (here with the CPU
oracle standing in for the device), one all_gather reassembles the results in global order."""
import os
import socket
import sys

import numpy as np
import pytest

from ihm2_amd.dist import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    for total in (1, 7, 64, 1000, 65536):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="2")
    import torch
    import torch.distributed as dist
    from conftest import make_ocp, sample_x0

    from ihm2_amd.dist import all_gather_blocks, shard
    from ihm2_amd.track import track_table
    from oracle import oracle as orc

    dist.init_process_group("gloo", rank=rank, world_size=world)
    track = track_table("fsds_competition_1")
    N = 10
    P = orc.OracleProblem(make_ocp(N=N, M=20).flatten().as_dict(track.s_ref, track.kappa_ref))
    x0_all = sample_x0(track, total, seed=3)
    x0 = shard(x0_all, world, rank)
    B = x0.shape[0]
    x = np.repeat(x0[:, None, :], N + 1, axis=1).copy(); u = np.zeros((B, N, 2))
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 10.0 * np.arange(N)[None] / N
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 10.0
    out = P.rti_step(x, u, x0, yref, yref_e, nthreads=2)
    u0 = all_gather_blocks(torch.from_numpy(u[:, 0].copy()), total)
    st = all_gather_blocks(torch.from_numpy(out["status"].astype(np.int32)), total)
    dist.barrier()
    if rank == 0:
        q.put((u0.numpy(), st.numpy()))
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(track):
    import torch.multiprocessing as mp
    from conftest import make_ocp, sample_x0

    from oracle import oracle as orc

    total, world = 11, 2          # uneven split: 6 + 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    u0, st = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the whole batch
    N = 10
    P = orc.OracleProblem(make_ocp(N=N, M=20).flatten().as_dict(track.s_ref, track.kappa_ref))
    x0 = sample_x0(track, total, seed=3)
    x = np.repeat(x0[:, None, :], N + 1, axis=1).copy(); u = np.zeros((total, N, 2))
    yref = np.zeros((total, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 10.0 * np.arange(N)[None] / N
    yref_e = np.zeros((total, 8)); yref_e[:, 0] = x0[:, 0] + 10.0
    out = P.rti_step(x, u, x0, yref, yref_e, nthreads=2)
    np.testing.assert_array_equal(st, out["status"])
    np.testing.assert_allclose(u0, u[:, 0], rtol=1e-12, atol=1e-12)


def test_global_monte_carlo_batch_is_the_concatenation_of_its_shards():
    """bench.py --config 3: every rank cuts rows [lo, hi) out of ONE global (x0, track_id); any world size gives the single-handle batch."""
    import bench
    from ihm2_amd.track import track_table

    plans = [track_table(t) for t in bench.ALL_TRACKS]
    for total in (7, 50, 1001):
        x0, tid = bench.monte_carlo_batch(plans, total)
        assert np.array_equal(tid, np.arange(total) % 7)
        for world in (1, 2, 3, 8):
            parts = [shard_bounds(total, world, r) for r in range(world)]
            assert np.array_equal(np.concatenate([x0[lo:hi] for lo, hi in parts]), x0)
            assert np.array_equal(np.concatenate([tid[lo:hi] for lo, hi in parts]), tid)
        for t, p in enumerate(plans):        # every instance starts on its own track
            assert np.all((x0[tid == t, 0] >= 0) & (x0[tid == t, 0] < p.lap_length))


def test_spawned_ranks_shard_a_non_divisible_batch_and_gather_in_global_order(tmp_path):
    """ihm2_amd.dist.spawn_ranks (what `bench.py --gpus N` does without a launcher) + RankContext over gloo: three ranks, 20 instances
    (7 + 7 + 6) on three tracks; the gathered result equals the single-process result of the whole batch."""
    import bench
    from conftest import make_ocp
    from ihm2_amd.dist import spawn_ranks
    from ihm2_amd.track import track_table
    from oracle import oracle as orc

    total, world = 20, 3
    out = tmp_path / "gathered.npz"
    rc = spawn_ranks(world, [os.path.join(ROOT, "tests", "_rank_worker.py"), "run", str(total), str(out)])
    assert rc == 0
    got = np.load(out)
    assert int(got["world"]) == world and float(got["slowest"]) == float(world) and float(got["count"]) == float(total)
    # the record of who answered: three ranks with their device ordinals (LOCAL_RANK), gathered over the carrier; gloo marks the run a rehearsal
    assert int(got["n_devices"]) == world and int(got["distinct"]) == world and bool(got["rehearsal"])
    plans = [track_table(t) for t in bench.ALL_TRACKS[:3]]
    x0, tid = bench.monte_carlo_batch(plans, total)
    N = 8
    s_ref = np.stack([p.s_ref for p in plans]); k_ref = np.stack([p.kappa_ref for p in plans])
    P = orc.OracleProblem(make_ocp(N=N, M=20).flatten().as_dict(s_ref, k_ref))
    x = np.repeat(x0[:, None, :], N + 1, axis=1).copy(); u = np.zeros((total, N, 2))
    yref = np.zeros((total, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 8.0 * np.arange(N)[None] / N
    yref_e = np.zeros((total, 8)); yref_e[:, 0] = x0[:, 0] + 8.0
    ref = P.rti_step(x, u, x0, yref, yref_e, track_id=tid, nthreads=2)
    np.testing.assert_array_equal(got["status"], ref["status"])
    np.testing.assert_allclose(got["u0"], u[:, 0], rtol=1e-12, atol=1e-12)


def test_spawned_ranks_report_a_rank_that_dies(tmp_path):
    from ihm2_amd.dist import spawn_ranks

    rc = spawn_ranks(2, [os.path.join(ROOT, "tests", "_rank_worker.py"), "fail", "4", str(tmp_path / "none.npz")])
    assert rc != 0


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess

    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_a_job_is_reported_as_n_gpus_only_if_n_distinct_devices_answered():
    """the bench line says ``n_gpus: N`` only if N distinct devices answered the gather of device identities over the
    communicator (``ihm2mpc_comm_info``); ranks that share a device are refused unless the run is a declared rehearsal."""
    from ihm2_amd.dist import pci_name, require_distinct_devices

    ids = [(0 << 24) | (b << 8) | 0 for b in (0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xE5, 0xF5)]
    assert require_distinct_devices(ids, 8) == 8
    assert pci_name(ids[2]) == "0000:65:00"
    with pytest.raises(SystemExit, match="7 distinct devices"):
        require_distinct_devices(ids[:7] + ids[:1], 8)                 # two ranks on one device
    with pytest.raises(SystemExit, match="answered"):
        require_distinct_devices(ids[:7], 8)                           # a rank is missing
    assert require_distinct_devices([ids[0]] * 2, 2, rehearsal=True) == 1
    # partitions of one package share the PCI triple and differ in the UUID tag (bits 48..62): distinct devices; the name shows both
    parts = [ids[2] | (t << 48) for t in (0x1234, 0x0777)]
    assert require_distinct_devices(parts, 2) == 2 and pci_name(parts[0]) == "0000:65:00 uuid~1234"
    with pytest.raises(SystemExit, match="1 distinct devices"):
        require_distinct_devices([parts[0]] * 2, 2)


def test_bench_refuses_one_gpu_workloads_on_several_ranks():
    """--config 2 and --live-options ignore the world size; with --gpus N > 1 every rank would run the whole workload on one device
    and print its own line.  Refused before any rank is started."""
    import subprocess

    for extra in (["--config", "2"], ["--live-options"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra, capture_output=True, text=True)
        assert r.returncode != 0 and "one-GPU" in r.stderr
