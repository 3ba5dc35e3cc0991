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


def exchange_bytes(payload: bytes | None, nbytes: int, world: int, rank: int, addr: str, port: int, timeout: float = 45.0) -> bytes:
    """Rank 0 hands ``payload`` (``nbytes`` long) to every other rank over TCP: the side channel of the RCCL unique id.  The wait is
    bounded well below the 120 s rendezvous of the fallback carrier (:meth:`RankContext.attach`): if rank 0 cannot bind the port it
    is in that rendezvous at once, and the others must give up here early enough to still meet it there."""
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

    def info(self):
        """``(ncclCommCount, PCI identity of every rank's device)`` -- gathered over the communicator itself."""
        from . import _lib

        n = C.c_int32(0); ids = (C.c_int64 * self.world)()
        _lib.check(self.lib.ihm2mpc_comm_info(self.solver._h, C.byref(n), ids))
        return int(n.value), [int(v) for v in ids]

    def free(self):
        from . import _lib

        if self.solver._h:          # a freed handle has taken its communicator with it (ihm2mpc_free)
            _lib.check(self.lib.ihm2mpc_comm_free(self.solver._h))


def pci_name(ident: int) -> str:
    """``domain:bus:device`` of a device identity of ``ihm2mpc_comm_info``."""
    tag = (ident >> 48) & 0x7FFF
    return f"{(ident >> 24) & 0xFFFF:04x}:{(ident >> 8) & 0xFFFF:02x}:{ident & 0xFF:02x}" + (f" uuid~{tag:04x}" if tag else "")


def require_distinct_devices(device_ids, world: int, rehearsal: bool = False) -> int:
    """A job of ``world`` ranks is reported as ``n_gpus = world`` only if ``world`` distinct devices answered the gather of identities;
    ``rehearsal`` (all ranks put on one device on purpose: ``bench.py --device D`` / gloo) waives it.  Returns the number of distinct devices."""
    distinct = len(set(int(d) for d in device_ids))
    if len(device_ids) != world:
        raise SystemExit(f"{len(device_ids)} ranks answered, the job has {world}")
    if distinct != world and not rehearsal:
        raise SystemExit(f"{world} ranks but {distinct} distinct devices ({', '.join(pci_name(int(d)) for d in device_ids)}): "
                         "not a {0}-GPU run -- refusing to report n_gpus = {0}".format(world))
    return distinct


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


class RankContext:
    """The ranks of one job and its three exchanges -- barrier, max over the ranks, final gather of ``(u0, status)`` -- over one of

    * ``"none"``   one rank: nothing to exchange;
    * ``"rccl"``   the C ABI's own RCCL communicator (:class:`NativeComm`, no PyTorch);
    * ``"torch"``  ``torch.distributed`` (backend ``nccl`` = RCCL on a GPU node, ``gloo`` to rehearse on one device or on the CPU).

    ``rank`` / ``local_rank`` / ``world`` come from the launcher's environment (``torch.distributed.run`` or :func:`spawn_ranks`).
    """

    def __init__(self, carrier: str = "auto", backend: str = "nccl", device: int | None = None):
        import os

        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device = self.local_rank if device is None else int(device)
        self.addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.port = int(os.environ.get("MASTER_PORT", "29500"))
        force = os.environ.get("IHM2_FORCE_DIST") == "1"          # rehearse the exchange path with one rank
        requested = carrier
        self.rehearsal = device is not None or backend != "nccl"   # ranks put on one device on purpose / no GPU collective at all
        if self.world == 1 and not force:
            carrier = "none"
        elif carrier == "auto":
            # the product's own RCCL communicator behind the C ABI (no PyTorch) wherever the ranks sit on distinct GPUs; torch.distributed
            # only where RCCL cannot run (gloo: the rehearsal of N ranks on one device / on the CPU)
            carrier = "rccl" if backend == "nccl" else "torch"
        self.carrier, self.backend = carrier, backend
        self.carrier_was_auto = carrier == "rccl" and requested == "auto"
        self.fallback = None          # why the default carrier was left, if it was (goes into the bench line)
        self.on_gpu = backend == "nccl"
        self._dist = None
        self._native = None
        if carrier == "torch":
            self._init_torch()

    def _init_torch(self, timeout_s: float | None = None) -> None:
        import datetime

        import torch
        import torch.distributed as dist

        kw = {} if timeout_s is None else {"timeout": datetime.timedelta(seconds=timeout_s)}
        if self.on_gpu:
            torch.cuda.set_device(self.device)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.device), **kw)
        else:
            dist.init_process_group(backend="gloo", **kw)
        self._dist = dist

    def bounds(self, total: int) -> tuple[int, int]:
        return shard_bounds(total, self.world, self.rank)

    def attach(self, solver, total: int) -> None:
        """``solver`` holds this rank's block of ``total`` instances (the RCCL carrier builds its communicator on the handle)."""
        self.total = int(total)
        if self.carrier == "rccl":
            if self._native is not None:
                self._native.free()
                self._native = None
            try:
                self._native = NativeComm(solver, total, self.world, self.rank, self.addr, self.port + 23)
            except Exception as e:
                # The default carrier could not be built (librccl not resolvable, the id channel's port taken, ncclCommInitRank refusing):
                # causes of that kind hit every rank of a node alike, so every rank lands here and the three exchanges -- barrier, max,
                # the final gather of 20 B per instance -- go over torch.distributed's gloo on HOST buffers, with the reason in the line.
                # Not over torch's nccl backend: torch imported after this library binds its c10d to the HIP runtime and the librccl
                # already mapped from /opt/rocm instead of its own bundled ones (measured: "No HIP GPUs are available", or a gather that
                # returns stale memory; tools/probes/torch_gather_diag.py).  An explicit `--carrier rccl` is not second-guessed, and a
                # failure on SOME ranks only ends in the bounded rendezvous timeout below instead of a result.
                if not self.carrier_was_auto:
                    raise
                self.fallback = f"{type(e).__name__}: {e}"
                self.carrier = "torch"
                self.on_gpu = False
                self._init_torch(timeout_s=120.0)

    def barrier(self, solver=None) -> None:
        if solver is not None:
            solver.synchronize()
        if self._native is not None:
            self._native.allreduce_max(0.0)
        if self._dist is not None:
            import torch

            if self.on_gpu:
                torch.cuda.synchronize()
            self._dist.barrier()
            if self.on_gpu:
                torch.cuda.synchronize()

    def max(self, value: float) -> float:
        if self._native is not None:
            return self._native.allreduce_max(value)
        if self._dist is not None:
            import torch

            t = torch.tensor([value], dtype=torch.float64, device="cuda" if self.on_gpu else "cpu")
            self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
            return float(t.item())
        return float(value)

    def sum(self, value: float) -> float:
        """Sum over the ranks of a host scalar (bookkeeping of the sharded configurations: solved instances, cars alive)."""
        if self._dist is not None:
            import torch

            t = torch.tensor([value], dtype=torch.float64, device="cuda" if self.on_gpu else "cpu")
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
            return float(t.item())
        if self._native is not None:        # the communicator's one reduction is max: rank r's value is the max of (value on r, -inf elsewhere)
            return float(np.sum([self._native.allreduce_max(value if self.rank == r else -np.inf) for r in range(self.world)]))
        return float(value)

    def gather_results(self, solver):
        """``(u0 (total, 2), status (total,))`` of every instance of the job in global order, on every rank."""
        if self._native is not None:
            return self._native.allgather_results()
        if self._dist is not None:
            import torch

            B = solver.B
            if self.on_gpu:
                u0_dev = torch.empty((B, 2), dtype=torch.float64, device="cuda")
                st_dev = torch.empty((B,), dtype=torch.int32, device="cuda")
                solver.get_u0_device(u0_dev.data_ptr()); solver.get_status_device(st_dev.data_ptr()); solver.synchronize()
            else:
                u0_dev = torch.from_numpy(solver.get_u0()); st_dev = torch.from_numpy(solver.get_status().copy())
            u0_all = all_gather_blocks(u0_dev, self.total)
            st_all = all_gather_blocks(st_dev, self.total)
            if self.on_gpu:
                torch.cuda.synchronize()
            return u0_all.cpu().numpy(), st_all.cpu().numpy()
        return solver.get_u0(), solver.get_status()

    def describe(self) -> str:
        return {"none": "none (one GPU)", "rccl": "RCCL all-gather of (u0, status) behind the C ABI (ihm2mpc_comm_*, no PyTorch)",
                "torch": f"torch.distributed ({'nccl = RCCL' if self.on_gpu else 'gloo, host buffers'}) all_gather of (u0, status)"}[self.carrier]

    def verify(self, solver) -> dict:
        """What actually answered: the communicator's rank count and every rank's device, gathered over the carrier itself.  Raises
        ``SystemExit`` when fewer distinct devices than ranks did (unless the run is a declared rehearsal).  The returned record goes
        into the bench line (``config.collective``)."""
        rec = {"carrier": self.describe(), "world": self.world}
        if self.fallback is not None:
            rec["fallback_from_rccl_carrier"] = self.fallback
        if self._native is not None:
            count, ids = self._native.info()
            rec.update(rccl_ranks=count, devices=[pci_name(i) for i in ids], distinct_devices=require_distinct_devices(ids, self.world, self.rehearsal))
            if count != self.world:
                raise SystemExit(f"the RCCL communicator spans {count} ranks, the job has {self.world}")
        elif self._dist is not None:
            import torch

            t = torch.tensor([float(self.device)], dtype=torch.float64, device="cuda" if self.on_gpu else "cpu")
            out = [torch.empty_like(t) for _ in range(self.world)]
            self._dist.all_gather(out, t)
            ids = [int(o.item()) for o in out]
            rec.update(devices=[f"local device {i}" for i in ids], distinct_devices=require_distinct_devices(ids, self.world, self.rehearsal))
            if self.on_gpu:
                rec["rccl_ranks"] = int(self._dist.get_world_size())
        else:
            rec.update(devices=[f"local device {self.device}"], distinct_devices=1)
        if self.rehearsal and self.world > 1:
            rec["rehearsal"] = "ranks share a device on purpose (--device / gloo): not a multi-GPU measurement"
        return rec

    def detach(self) -> None:
        """Give the solver's communicator back BEFORE the solver is freed (``ihm2mpc_free`` takes the communicator with the handle)."""
        if self._native is not None:
            self._native.allreduce_max(0.0)
            self._native.free(); self._native = None

    def close(self) -> None:
        self.detach()
        if self._dist is not None:
            self._dist.barrier()
            self._dist.destroy_process_group(); self._dist = None

    def abort(self) -> None:
        """Leave without synchronising: this rank is unwinding from an error while the others sit in some collective or will never reach a
        matching one -- a barrier here would turn a failure into a hang.  The launcher stops the peers once this rank has exited."""
        self._native = None
        if self._dist is not None:
            try:
                self._dist.destroy_process_group()
            except Exception:
                pass
            self._dist = None


def spawn_ranks(n: int, argv: list[str], extra_env: dict | None = None) -> int:
    """Start ``n`` copies of ``argv`` -- one process per GPU, ``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE`` / ``MASTER_*`` set as
    ``torch.distributed.run`` would -- and wait for them.  The caller has not touched the GPU (children must not be forked from a
    process that has).  Returns 0 if every rank did; otherwise the first failing rank's exit code, after stopping the others."""
    import os
    import subprocess
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable] + argv, env=env))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:          # a rank that died leaves the others waiting in a collective: stop exactly those we started
                    procs[q].terminate()
        time.sleep(0.05)
    return rc
