"""Diagnostic: the soft a_lat table (`k_qp_wave<10,4,2,1>`) and a four-soft-per-lane table without the row (`<10,4,1,1>`) in the default and the ilp build."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_ocp, sample_x0
from ihm2_amd import _lib as L
from ihm2_amd.track import track_table
from ihm2_amd.solver import BatchedOcpSolver
track = track_table("fsds_competition_1")
N = 40
def ocp_for(kind):
    ocp = make_ocp()
    c = ocp.constraints
    if kind == "alat":
        ocp.model.con_h_expr = "track+a_lat"
        c.lh = np.array([-1e3, -1e3, -2.5]); c.uh = np.array([0.0, 0.0, 2.5]); c.lh_e = c.lh[:2]; c.uh_e = c.uh[:2]
        c.idxsh, c.idxsh_e = np.arange(3), np.arange(2)
        ocp.cost.zl = ocp.cost.zu = ocp.cost.Zl = ocp.cost.Zu = np.full(3, 100.0)
    else:       # soft track rows + a soft n box: four soft sides per lane without the row
        ocp.model.con_h_expr = "track"
        c.lh = c.lh_e = np.array([-1e3, -1e3]); c.uh = c.uh_e = np.array([0.0, 0.0])
        c.idxsbx = np.array([0]); c.idxsh, c.idxsh_e = np.arange(2), np.arange(2)
        ocp.cost.zl = ocp.cost.zu = ocp.cost.Zl = ocp.cost.Zu = np.full(3, 100.0)
    ocp.cost.zl_e = ocp.cost.zu_e = ocp.cost.Zl_e = ocp.cost.Zu_e = np.full(2, 100.0)
    return ocp
BASE = os.path.dirname(L.LIB_PATH)
for build in ["default", "ilp"] + sys.argv[1:]:     # further arguments: paths of diagnostic builds
    if build != "default":
        L._lib, L.LIB_PATH = None, (os.path.join(BASE, "libihm2mpc_ilp.so") if build == "ilp" else build)
    for kind in ("alat", "path4"):
        B = 256
        s = BatchedOcpSolver(ocp_for(kind), B, track.s_ref, track.kappa_ref, track_widths=np.array([[1.6, 1.5]]))
        x0 = sample_x0(track, B, seed=4321); x0[:, 3] = np.linspace(8, 14, B)
        s.set_x0(x0); s.init_guess()
        yref = np.zeros((B, N, 12)); yref[:, :, 0] = x0[:, 0:1] + 60.0 * np.arange(N)[None] / N; yref[:, :, 3] = 15.0
        yref_e = np.zeros((B, 8)); yref_e[:, 0] = x0[:, 0] + 60.0; yref_e[:, 3] = 15.0
        s.set_yref(yref); s.set_yref_e(yref_e); s.set_multipliers(None, None)
        st = s.solve()
        it = s.get_qp_iter()
        print(build, kind, "status", np.bincount(st, minlength=5), "iters", np.bincount(it)[:40], "qp_res max", np.nanmax(s.get_qp_residuals(), axis=0), flush=True)
        s.free()
