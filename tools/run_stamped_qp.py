"""Runs the cycle-stamped diagnostic build of the QP kernel (tools/make_stamped_qp.py) on the bench workload and prints the
per-section cycle counts of a few instances; never part of the product.

    python tools/make_stamped_qp.py && python tools/run_stamped_qp.py        # on the GPU box
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ihm2_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "scratch", "csrc_dbg", "libihm2mpc_dbg.so")
import bench  # noqa: E402
from ihm2_amd.solver import BatchedOcpSolver  # noqa: E402

if __name__ == "__main__":
    B = 1024
    ocp, track = bench.build_problem(B)
    s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref)
    s.set_x0(bench.sample_x0(track, B, seed=20240607)); s.init_guess()
    for _ in range(12):          # the stamped launcher prints the sections of its first launches to stderr
        s.step(bench.S_TARGET, model=0, M_sim=bench.M_SUB)
        s.synchronize()          # the launcher reads the stamps of the PREVIOUS launch without waiting for the stream
    s.synchronize()
    print("qp_iter mean", s.get_qp_iter().mean(), "timings", s.get_timings())
