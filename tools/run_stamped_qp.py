"""Runs the cycle-stamped diagnostic build of the QP kernel (tools/make_stamped_qp.py) on the bench workload and prints the
per-section cycle counts of a few instances; never part of the product.

    python tools/make_stamped_qp.py && python tools/run_stamped_qp.py        # on the GPU box
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ihm2_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.environ.get("IHM2_DBG_LIB", os.path.join(ROOT, "scratch", "csrc_dbg", "libihm2mpc_dbg.so"))
import bench  # noqa: E402
from ihm2_amd.solver import BatchedOcpSolver  # noqa: E402

if __name__ == "__main__":
    B = 1024
    ocp, track = bench.build_problem(B)
    kw = {}
    if "--soft" in sys.argv:      # soft nonlinear track rows 100 / 100: the <8,3,1,1> instantiation (build with STAMP_SOFT=1)
        import numpy as np
        ocp.model.con_h_expr = "track"
        c = ocp.constraints
        c.lh = c.lh_e = np.array([-1e3, -1e3]); c.uh = c.uh_e = np.array([0.0, 0.0])
        c.idxsh, c.idxsh_e = np.arange(2), np.arange(2)
        ocp.cost.zl = ocp.cost.zu = ocp.cost.Zl = ocp.cost.Zu = np.full(2, 100.0)
        ocp.cost.zl_e = ocp.cost.zu_e = ocp.cost.Zl_e = ocp.cost.Zu_e = np.full(2, 100.0)
        kw["track_widths"] = np.array([[track.right_widths.min(), track.left_widths.min()]])
    s = BatchedOcpSolver(ocp, B, track.s_ref, track.kappa_ref, **kw)
    s.set_x0(bench.sample_x0(track, B, seed=20240607)); s.init_guess()
    for _ in range(12):          # the stamped launcher prints the sections of its first launches to stderr
        s.step(bench.S_TARGET, model=0, M_sim=bench.M_SUB)
        s.synchronize()          # the launcher reads the stamps of the PREVIOUS launch without waiting for the stream
    s.synchronize()
    print("qp_iter mean", s.get_qp_iter().mean(), "timings", s.get_timings())
