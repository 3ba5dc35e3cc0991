"""The native (PyTorch-free) gather of the multi-GPU path: sharding arithmetic and the RCCL-id side channel on the CPU; the
communicator itself (world size 1, the only size a one-GPU box offers) on the GPU."""
import threading

import numpy as np
import pytest

from conftest import make_ocp, sample_x0


def test_shard_sizes_cover_the_batch_like_a_single_handle():
    from ihm2_amd.dist import shard, shard_bounds, shard_sizes

    for total, world in ((65536, 8), (4096, 8), (1000, 7), (5, 8), (1, 1)):
        sizes = shard_sizes(total, world)
        assert sizes.sum() == total and sizes.max() - sizes.min() <= 1
        x = np.arange(total * 3).reshape(total, 3)
        parts = [shard(x, world, r) for r in range(world)]
        assert [p.shape[0] for p in parts] == list(sizes)
        np.testing.assert_array_equal(np.concatenate(parts), x)          # rank order = global order: the gather needs no permutation
        off = 0
        for r in range(world):
            assert shard_bounds(total, world, r) == (off, off + int(sizes[r]))
            off += int(sizes[r])


def test_the_id_channel_delivers_to_every_rank():
    from ihm2_amd.dist import exchange_bytes

    world, port = 4, 29611
    payload = bytes(range(128))
    got = [None] * world

    def run(rank):
        got[rank] = exchange_bytes(payload if rank == 0 else None, 128, world, rank, "127.0.0.1", port, timeout=30.0)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert all(g == payload for g in got)


@pytest.mark.gpu
def test_native_rccl_communicator_on_one_gpu(track):
    """ihm2mpc_comm_* and ihm2mpc_group_* with one rank / one device: the collective returns this rank's own results (on 8 GPUs the
    same calls gather 8 blocks; the driver's scaling run exercises that)."""
    from ihm2_amd.dist import DeviceGroup, NativeComm
    from ihm2_amd.solver import BatchedOcpSolver

    B = 70
    s = BatchedOcpSolver(make_ocp(), B, track.s_ref, track.kappa_ref)
    s.set_x0(sample_x0(track, B, seed=4)); s.init_guess(); s.prepare_step(40.0); st = s.solve()
    comm = NativeComm(s, B, 1, 0)
    u0, status = comm.allgather_results()
    np.testing.assert_array_equal(status, st)
    np.testing.assert_array_equal(u0, s.get_u0())
    assert comm.allreduce_max(3.25) == 3.25
    count, ids = comm.info()                       # ncclCommCount and the PCI identity of every rank's device, gathered over the communicator
    assert count == 1 and len(ids) == 1 and ids[0] >= 0
    with pytest.raises(Exception):
        NativeComm(s, B, 1, 0)                     # one communicator per handle
    comm.free()
    g = DeviceGroup([s])
    u0g, stg = g.allgather_results()
    np.testing.assert_array_equal(u0g, u0); np.testing.assert_array_equal(stg, st)
    g.free()
    s.free()


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--config", "3", "--batch", "192", "--steps", "2", "--warmup", "1"],
                                   ["--config", "4", "--batch", "64", "--steps", "3"],
                                   ["--batch", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]])
def test_bench_over_the_native_communicator_with_one_forced_rank(extra):
    """`bench.py --config 3` over the C ABI's own RCCL communicator used the communicator after `solver.free()` had destroyed it
    (no JSON line).  One rank with the exchange path forced (IHM2_FORCE_DIST=1): the line is printed and carries the communicator's record."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IHM2_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--carrier", "rccl"] + extra, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    col = line["config"]["collective"]
    assert col["rccl_ranks"] == 1 and col["distinct_devices"] == 1 and len(col["devices"]) == 1 and "RCCL" in col["carrier"]
    assert line["n_gpus"] == 1 and line["value"] > 0 and "gather_ms" in line


def _one_forced_rank(monkeypatch, port):
    for k, v in dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IHM2_FORCE_DIST="1").items():
        monkeypatch.setenv(k, v)


def test_default_carrier_falls_back_when_its_communicator_cannot_be_built(monkeypatch):
    """`--carrier auto` = the C ABI's own RCCL communicator; if that cannot be BUILT (librccl not resolvable, the id channel's port taken --
    the same on every rank of a node) the job goes on over torch.distributed and says why; an explicit `--carrier rccl` raises."""
    from ihm2_amd import dist

    _one_forced_rank(monkeypatch, 29633)

    def broken(*a, **k):
        raise RuntimeError("librccl.so: cannot open shared object file")

    started = []
    monkeypatch.setattr(dist, "NativeComm", broken)
    monkeypatch.setattr(dist.RankContext, "_init_torch", lambda self, timeout_s=None: started.append(timeout_s))
    ctx = dist.RankContext(carrier="auto", backend="nccl")
    assert ctx.carrier == "rccl" and ctx.fallback is None
    ctx.attach(object(), 8)
    assert ctx.carrier == "torch" and started == [120.0] and "librccl" in ctx.fallback
    assert "gloo" in ctx.describe() and not ctx.on_gpu          # host buffers: torch's nccl backend must not be started after this library
    explicit = dist.RankContext(carrier="rccl", backend="nccl")
    with pytest.raises(RuntimeError):
        explicit.attach(object(), 8)
    assert explicit.carrier == "rccl" and started == [120.0]


@pytest.mark.gpu
def test_fallback_carrier_gathers_the_same_results(track, monkeypatch):
    """The fallback for real on one forced rank, in a process that has the product's HIP runtime and RCCL mapped already: the native
    communicator refused, torch.distributed's gloo takes the three exchanges on host buffers."""
    from ihm2_amd import dist
    from ihm2_amd.solver import BatchedOcpSolver

    _one_forced_rank(monkeypatch, 29635)

    def broken(*a, **k):
        raise OSError("address already in use")

    monkeypatch.setattr(dist, "NativeComm", broken)
    B = 70
    s = BatchedOcpSolver(make_ocp(), B, track.s_ref, track.kappa_ref)
    s.set_x0(sample_x0(track, B, seed=4)); s.init_guess(); s.prepare_step(40.0); st = s.solve()
    ctx = dist.RankContext(carrier="auto", backend="nccl")
    try:
        ctx.attach(s, B)
        assert ctx.carrier == "torch" and "address already in use" in ctx.fallback
        ctx.barrier(s)
        assert ctx.max(3.25) == 3.25 and ctx.sum(2.0) == 2.0
        u0, status = ctx.gather_results(s)
        np.testing.assert_array_equal(status, st)
        np.testing.assert_array_equal(u0, s.get_u0())
        rec = ctx.verify(s)
        assert rec["distinct_devices"] == 1 and "gloo" in rec["carrier"] and "address already in use" in rec["fallback_from_rccl_carrier"]
    finally:
        ctx.close()
        s.free()
