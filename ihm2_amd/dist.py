"""Multi-GPU sharding of a batch of independent MPC instances.

The path shards trivially: instances do not interact, so GPU g owns the contiguous block
``[g*B/G, (g+1)*B/G)`` of the global batch, track tables are replicated, and there is no data-path
collective.  The only communication is ONE gather of the result block at the end:

* ``NativeComm`` / ``DeviceGroup`` -- RCCL over xGMI behind the C ABI (``ihm2mpc_comm_*`` / ``ihm2mpc_group_*``, csrc/comm.hip),
  no PyTorch: one process per GPU with the 128-byte RCCL id carried over a TCP socket on ``MASTER_ADDR``, or one process
  driving one handle per device;
* ``all_gather_blocks`` -- the same gather over ``torch.distributed`` (backend ``nccl`` = RCCL on the GPU node, ``gloo`` in the
  CPU tests); ``torch`` is plumbing there (process group, device buffers), imported lazily, never in the solver arithmetic.
"""
from __future__ import annotations

import ctypes as C
import socket
import time

import numpy as np


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block split; the first ``total % world`` ranks get one extra instance."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} out of range for world size {world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(array: np.ndarray, world: int, rank: int) -> np.ndarray:
    lo, hi = shard_bounds(array.shape[0], world, rank)
    return array[lo:hi]


def all_gather_blocks(local, total: int, group=None):
    """Gather per-rank result blocks (first axis = local instances) into the global array on every rank.

    ``local`` is a ``torch.Tensor`` living where the process group communicates (GPU for nccl, CPU for
    gloo).  Uneven shards are padded to the largest block for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
    nmax = max(sizes)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local block does not match this rank's shard")
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)


def shard_sizes(total: int, world: int) -> np.ndarray:
    return np.array([shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)], dtype=np.int32)


def exchange_bytes(payload: bytes | None, nbytes: int, world: int, rank: int, addr: str, port: int, timeout: float = 120.0) -> bytes:
    """Rank 0 hands ``payload`` (``nbytes`` long) to every other rank over TCP: the side channel of the RCCL unique id."""
    if world == 1:
        return payload
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port)); srv.listen(world)
        srv.settimeout(timeout)
        for _ in range(world - 1):
            conn, _ = srv.accept()
            conn.sendall(payload); conn.close()
        srv.close()
        return payload
    t_end = time.time() + timeout
    while True:
        try:
            c = socket.create_connection((addr, port), timeout=5.0)
            break
        except OSError:
            if time.time() > t_end:
                raise
            time.sleep(0.1)
    buf = b""
    while len(buf) < nbytes:
        chunk = c.recv(nbytes - len(buf))
        if not chunk:
            raise ConnectionError("the id channel closed early")
        buf += chunk
    c.close()
    return buf


class NativeComm:
    """RCCL communicator of a one-process-per-GPU job behind the C ABI (no PyTorch).  ``solver``: this rank's
    ``BatchedOcpSolver`` (its batch = this rank's block of the ``total`` instances)."""

    def __init__(self, solver, total: int, world: int, rank: int, addr: str = "127.0.0.1", port: int = 29523):
        from . import _lib

        self.lib, self.solver, self.world, self.rank, self.total = _lib.load(), solver, world, rank, int(total)
        self.sizes = shard_sizes(total, world)
        if int(self.sizes[rank]) != solver.B:
            raise ValueError(f"rank {rank} owns {int(self.sizes[rank])} of {total} instances, the solver holds {solver.B}")
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            _lib.check(self.lib.ihm2mpc_comm_unique_id(ident))
        raw = exchange_bytes(bytes(ident) if rank == 0 else None, 128, world, rank, addr, port)
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        _lib.check(self.lib.ihm2mpc_comm_init(solver._h, world, rank, ident, self.sizes.ctypes.data_as(_lib.c_int32_p)))

    def allgather_results(self):
        """(u0 of every instance (total, 2), status (total,)) on every rank, in rank order."""
        from . import _lib

        u0 = np.empty((self.total, 2)); st = np.empty(self.total, dtype=np.int32)
        _lib.check(self.lib.ihm2mpc_comm_allgather_results(self.solver._h, u0.ctypes.data_as(_lib.c_double_p), st.ctypes.data_as(_lib.c_int32_p)))
        return u0, st

    def allreduce_max(self, value: float) -> float:
        from . import _lib

        v = C.c_double(value)
        _lib.check(self.lib.ihm2mpc_comm_allreduce_max(self.solver._h, C.byref(v)))
        return float(v.value)

    def free(self):
        from . import _lib

        _lib.check(self.lib.ihm2mpc_comm_free(self.solver._h))


class DeviceGroup:
    """One process driving one ``BatchedOcpSolver`` per device: ``ncclCommInitAll`` behind ``ihm2mpc_group_create``."""

    def __init__(self, solvers):
        from . import _lib

        self.lib, self.solvers = _lib.load(), list(solvers)
        arr = (C.c_void_p * len(self.solvers))(*[s._h for s in self.solvers])
        self._g = C.c_void_p()
        _lib.check(self.lib.ihm2mpc_group_create(arr, len(self.solvers), C.byref(self._g)))
        self.total = sum(s.B for s in self.solvers)

    def allgather_results(self):
        from . import _lib

        u0 = np.empty((self.total, 2)); st = np.empty(self.total, dtype=np.int32)
        _lib.check(self.lib.ihm2mpc_group_allgather_results(self._g, u0.ctypes.data_as(_lib.c_double_p), st.ctypes.data_as(_lib.c_int32_p)))
        return u0, st

    def free(self):
        from . import _lib

        if self._g:
            _lib.check(self.lib.ihm2mpc_group_free(self._g)); self._g = C.c_void_p()
