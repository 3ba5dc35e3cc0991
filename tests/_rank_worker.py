"""Child process of tests/test_dist_gloo.py::test_spawned_ranks_*: one rank of a job started by ihm2_amd.dist.spawn_ranks.
The CPU oracle stands in for the device (test infrastructure); the exchanges go through RankContext over gloo."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


class OracleShard:
    """What RankContext needs of a solver: the block's batch size, u0, status."""

    def __init__(self, u0, status):
        self.B, self._u0, self._st = u0.shape[0], u0, status

    def synchronize(self):
        pass

    def get_u0(self):
        return self._u0

    def get_status(self):
        return self._st


def main():
    mode, total, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    if mode == "fail" and os.environ["RANK"] == "1":
        sys.exit(3)                                    # a rank that dies: spawn_ranks must stop the others and report it
    os.environ["OMP_NUM_THREADS"] = "2"
    import bench
    from conftest import make_ocp
    from ihm2_amd.dist import RankContext
    from ihm2_amd.track import track_table
    from oracle import oracle as orc

    ranks = RankContext(carrier="torch", backend="gloo")
    if mode == "fail":
        import time
        ranks.barrier()                                # never completes: rank 1 is gone; the parent terminates this process
        time.sleep(60)
    plans = [track_table(t) for t in bench.ALL_TRACKS[:3]]
    x0_all, tid_all = bench.monte_carlo_batch(plans, total)
    lo, hi = ranks.bounds(total)
    x0, tid = np.ascontiguousarray(x0_all[lo:hi]), np.ascontiguousarray(tid_all[lo:hi])
    N = 8
    s_ref = np.stack([p.s_ref for p in plans]); k_ref = np.stack([p.kappa_ref for p in plans])
    P = orc.OracleProblem(make_ocp(N=N, M=20).flatten().as_dict(s_ref, k_ref))
    B = hi - lo
    x = np.repeat(x0[:, None, :], N + 1, axis=1).copy(); u = np.zeros((B, N, 2))
    yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 8.0 * np.arange(N)[None] / N
    yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 8.0
    out = P.rti_step(x, u, x0, yref, yref_e, track_id=tid, nthreads=2)
    shard = OracleShard(u[:, 0].copy(), out["status"].astype(np.int32))
    ranks.attach(shard, total)
    ranks.barrier(shard)
    slowest = ranks.max(float(ranks.rank + 1))
    count = ranks.sum(float(B))
    u0_all, st_all = ranks.gather_results(shard)
    rec = ranks.verify(shard)                          # gloo = a declared rehearsal: ranks may share a device, and the record says so
    if ranks.rank == 0:
        np.savez(out_path, u0=u0_all, status=st_all, slowest=slowest, count=count, world=ranks.world, distinct=rec["distinct_devices"],
                 rehearsal="rehearsal" in rec, n_devices=len(rec["devices"]))
    ranks.close()


if __name__ == "__main__":
    main()
