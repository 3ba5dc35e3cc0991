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
    with pytest.raises(Exception):
        NativeComm(s, B, 1, 0)                     # one communicator per handle
    comm.free()
    g = DeviceGroup([s])
    u0g, stg = g.allgather_results()
    np.testing.assert_array_equal(u0g, u0); np.testing.assert_array_equal(stg, st)
    g.free()
    s.free()
